"""Decode GEMM shapes at several batch sizes (M = 16 / 32 / 48 / 64), graph-replayed back to back over rotating weights, for the automatic
and every forced block-row count R (debug bits 20-23; TB/s of weight bytes in brackets).  rows_min_mb is raised so that every M stays
on the skinny kernel (tools/rows_kernel_check.py times the rows kernel against it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import _lib, ops
L = _lib.lib()
L.mc_gemm_set_option(b"rows_min_mb", 5)

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

for M in (16, 32, 48, 64):
    for (N, K, name, kw) in ((4096, 4096, "o_proj", dict(res=True)), (4096, 11008, "down_proj", dict(res=True)), (12288, 4096, "qkv", dict(eps=1e-5)),
                             (22016, 4096, "gate|up", dict(eps=1e-5, sw=True))):
        ws = [ops.pack_weight(torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02) for _ in range(8)]
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        sw = kw.get("sw", False)
        out = torch.randn(M, N // 2 if sw else N, device="cuda", dtype=torch.bfloat16)
        i = [0]
        def f():
            i[0] = (i[0] + 1) % 8
            ops.linear_ex(x, ws[i[0]], residual=out if kw.get("res") else None, out=out, swiglu=sw, rms_eps=kw.get("eps", 0.0))
        res = []
        sweep = [("auto", 0)] + [(f"R={r}", r << 20) for r in (1, 2, 3, 4, 6, 8) if r * ((M + 15) // 16) <= 16 and not (sw and r % 2)]
        for label, dbg in sweep:
            L.mc_gemm_debug(dbg)
            res.append((label, timeit(f)))
        L.mc_gemm_debug(0)
        by = N * K * 2
        print(f"M={M:3d} {name:10s}: " + "  ".join(f"{l} {t*1e6:6.2f} us ({by/t/1e12:4.2f})" for l, t in res), flush=True)
