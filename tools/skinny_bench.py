import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import ops

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n // 20): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n // 20 * 20)

M = 16
for (N, K, name) in ((4096, 4096, "o_proj"), (4096, 11008, "down_proj"), (12288, 4096, "qkv")):
    # rotate through several weight copies so that nothing is served from the Infinity Cache
    ws = [ops.pack_weight(torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02) for _ in range(8)]
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    h = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    rs = torch.ones(M, device="cuda")
    i = [0]
    def nosplit():
        i[0] = (i[0] + 1) % 8
        ops.linear_ex(x, ws[i[0]], residual=h, out=h)
    part = torch.empty(4, M, N, device="cuda", dtype=torch.float32)
    def split():
        i[0] = (i[0] + 1) % 8
        ops.linear_ex(x, ws[i[0]], split_k=4, out=part)
        ops.residual_rms(h, part, 1e-5)
    by = N * K * 2
    for nm, fn in (("no split, residual epilogue", nosplit), ("split-4 + residual_rms", split)):
        t = timeit(fn)
        print(f"{name:10s} N={N} K={K} {nm:30s}: {t*1e6:7.2f} us  {by/t/1e12:5.2f} TB/s")

# lm_head and gate|up shapes through the default path (R chosen by the library)
for (N, K, name, sw) in ((32000, 4096, "lm_head", False), (22016, 4096, "gate|up swiglu", True)):
    ws = [ops.pack_weight(torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02) for _ in range(4)]
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    i = [0]
    for eps in (0.0, 1e-5):
        out = torch.empty(M, N // 2 if sw else N, device="cuda", dtype=torch.float32 if not sw else torch.bfloat16)
        def f():
            i[0] = (i[0] + 1) % 4
            ops.linear_ex(x, ws[i[0]], swiglu=sw, out=out, out_f32=not sw, rms_eps=eps)
        t = timeit(f)
        print(f"{name:14s} N={N} K={K} rms_eps={eps}: {t*1e6:7.2f} us  {N*K*2/t/1e12:5.2f} TB/s")
