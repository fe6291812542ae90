import sys, time, torch
sys.path.insert(0, "/root/repo")
from modelcompose_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
M,N,K=98688,4096,1024
x=torch.randn(M,K,device="cuda",dtype=torch.bfloat16)
w=ops.pack_weight(torch.randn(N,K,device="cuda",dtype=torch.bfloat16)*0.03, torch.randn(N,device="cuda",dtype=torch.bfloat16))
out=torch.empty(M,N,device="cuda",dtype=torch.bfloat16)
for act in ("none","quick_gelu","gelu","silu"):
    dt=t(lambda: ops.linear(x,w,act=act,out=out))
    print(act, f"{dt*1e6:.0f} us  {2*M*N*K/dt/1e12:.0f} TFLOP/s")
M2,N2,K2=98688,1024,4096
x2=torch.randn(M2,K2,device="cuda",dtype=torch.bfloat16)
w2=ops.pack_weight(torch.randn(N2,K2,device="cuda",dtype=torch.bfloat16)*0.03, torch.randn(N2,device="cuda",dtype=torch.bfloat16))
r=torch.randn(M2,N2,device="cuda",dtype=torch.bfloat16); o2=torch.empty_like(r)
dt=t(lambda: ops.linear(x2,w2,residual=r,out=o2)); print("fc2", f"{dt*1e6:.0f} us  {2*M2*N2*K2/dt/1e12:.0f} TFLOP/s")
x3=torch.randn(M,K,device="cuda",dtype=torch.bfloat16); w3=ops.pack_weight(torch.randn(3072,K,device="cuda",dtype=torch.bfloat16)*0.03, torch.randn(3072,device="cuda",dtype=torch.bfloat16)); o3=torch.empty(M,3072,device="cuda",dtype=torch.bfloat16)
dt=t(lambda: ops.linear(x3,w3,out=o3)); print("qkv", f"{dt*1e6:.0f} us  {2*M*3072*K/dt/1e12:.0f} TFLOP/s")
w4=ops.pack_weight(torch.randn(1024,K,device="cuda",dtype=torch.bfloat16)*0.03, torch.randn(1024,device="cuda",dtype=torch.bfloat16)); o4=torch.empty(M,1024,device="cuda",dtype=torch.bfloat16)
dt=t(lambda: ops.linear(x3,w4,residual=o4,out=o4)); print("out", f"{dt*1e6:.0f} us  {2*M*1024*K/dt/1e12:.0f} TFLOP/s")
M5, N5, K5 = 44656, 22016, 4096
x5 = torch.randn(M5, K5, device="cuda", dtype=torch.bfloat16)
w5 = ops.pack_weight(torch.randn(N5, K5, device="cuda", dtype=torch.bfloat16) * 0.02)
rs = torch.rand(M5, device="cuda") + 0.5
o5 = torch.empty(M5, N5 // 2, device="cuda", dtype=torch.bfloat16)
dt = t(lambda: ops.linear_ex(x5, w5, row_scale=rs, swiglu=True, out=o5), 5); print("gate|up swiglu", f"{dt*1e6:.0f} us  {2*M5*N5*K5/dt/1e12:.0f} TFLOP/s")
o6 = torch.empty(M5, N5, device="cuda", dtype=torch.bfloat16)
dt = t(lambda: ops.linear_ex(x5, w5, row_scale=rs, out=o6), 5); print("gate|up plain", f"{dt*1e6:.0f} us  {2*M5*N5*K5/dt/1e12:.0f} TFLOP/s")
M7, N7, K7 = 44656, 4096, 4096
x7 = torch.randn(M7, K7, device="cuda", dtype=torch.bfloat16)
w7 = ops.pack_weight(torch.randn(N7, K7, device="cuda", dtype=torch.bfloat16) * 0.02)
h7 = torch.randn(M7, N7, device="cuda", dtype=torch.bfloat16)
dt = t(lambda: ops.linear_ex(x7, w7, residual=h7, out=h7), 10); print("o_proj residual", f"{dt*1e6:.0f} us  {2*M7*N7*K7/dt/1e12:.0f} TFLOP/s")
dt = t(lambda: ops.linear_ex(x7, w7, out=h7), 10); print("o_proj plain", f"{dt*1e6:.0f} us  {2*M7*N7*K7/dt/1e12:.0f} TFLOP/s")
