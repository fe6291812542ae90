import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from modelcompose_amd import ops
M=2728
def run(P,Q,n,label):
    a=torch.randn(M,n*P,device="cuda").to(torch.bfloat16); b=torch.randn(M,n*Q,device="cuda").to(torch.bfloat16)
    al=[a[:,i*P:(i+1)*P] for i in range(n)]; bl=[b[:,i*Q:(i+1)*Q] for i in range(n)]
    outs=[torch.empty(P,Q,device="cuda") for _ in range(n)]
    for _ in range(3): ops.gemm_tn(al,bl,outs)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(50): ops.gemm_tn(al,bl,outs)
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/50
    by=M*n*(P+Q)*2+n*P*Q*4
    print(f"{label:28s} P={P:6d} Q={Q:6d} n={n}: {t*1e6:7.1f} us  {by/t/1e12:5.2f} TB/s unique")
run(4096,256,3,"attn_in dB (q,k,v)")
run(768,4096,1,"attn_in dA")
run(4096,256,1,"attn_out dB")
run(256,4096,1,"attn_out dA")
run(11008,256,2,"mlp_in dB (gate,up)")
run(512,4096,1,"mlp_in dA")
run(4096,256,1,"mlp_out dB")
run(256,11008,1,"mlp_out dA")
