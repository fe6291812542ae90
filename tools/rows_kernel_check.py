"""gemm_rows_kernel against the skinny kernel (debug bit 19 forces the old path) and an fp32 torch product, for the decode shapes at
M = 17 .. 64 and every forced split S; then timings (graph-replayed over rotating weights)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import _lib, ops
L = _lib.lib()
OLD = 1 << 19

def run(x, w, dbg, **kw):
    L.mc_gemm_debug(dbg)
    y = ops.linear_ex(x, w, **kw)
    L.mc_gemm_debug(0)
    return y

torch.manual_seed(0)
bad = 0
for M in (17, 32, 33, 48, 64):
    for (N, K, name) in ((4096, 4096, "o"), (4096, 11008, "down"), (12288, 4096, "qkv"), (22016, 4096, "gate_up"), (32000, 4096, "head"), (1000, 256, "small")):
        wd = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
        w = ops.pack_weight(wd)
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        res = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
        variants = [("plain", {}), ("res", dict(residual=res)), ("rms", dict(rms_eps=1e-5)), ("f32", dict(out_f32=True))]
        if N % 32 == 0:
            variants.append(("swiglu+rms", dict(swiglu=True, rms_eps=1e-5)))
        for vn, kw in variants:
            ref = run(x, w, OLD, **kw).float()
            for S in (0, 1, 2, 3, 4, 8):
                for rw2 in (0, 1, 2):
                    y = run(x, w, (S << 24) | ((rw2 & 1) << 28) | ((rw2 >> 1) << 30), **kw).float()
                    err = (y - ref).abs().max().item() / ref.abs().max().item()
                    if not err < 8e-3:
                        bad += 1
                        print(f"MISMATCH M={M} {name} {vn} S={S} rw2={rw2}: {err:.3e}")
        # fp32 truth for the plain case
        truth = x.float() @ wd.float().t()
        y = run(x, w, 0).float()
        e_new = (y - truth).abs().max().item() / truth.abs().max().item()
        e_old = (run(x, w, OLD).float() - truth).abs().max().item() / truth.abs().max().item()
        print(f"M={M:2d} {name:8s}: rel err vs fp32  rows {e_new:.2e}  skinny {e_old:.2e}", flush=True)
# repeated launches leave the counters at zero: same bits every time
x = torch.randn(48, 4096, device="cuda", dtype=torch.bfloat16)
w = ops.pack_weight(torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16) * 0.02)
y0 = ops.linear_ex(x, w)
for _ in range(50):
    assert torch.equal(ops.linear_ex(x, w), y0)
print("mismatches:", bad)
if "--time" not in sys.argv:
    sys.exit(1 if bad else 0)

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

L.mc_gemm_reserve_rows(torch.cuda.current_stream().cuda_stream)
for M in (16, 32, 48, 64) if "--all" in sys.argv else ((48,) if "--m48" in sys.argv else (32, 48)):
    for (N, K, name, kw) in ((4096, 4096, "o_proj", dict(res=True)), (4096, 11008, "down_proj", dict(res=True)), (12288, 4096, "qkv", dict(eps=1e-5)),
                             (22016, 4096, "gate|up", dict(eps=1e-5, sw=True)), (32000, 4096, "lm_head", dict(f32=True))):
        ws = [ops.pack_weight(torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02) for _ in range(8)]
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        sw = kw.get("sw", False)
        out = torch.randn(M, N // 2 if sw else N, device="cuda", dtype=torch.float32 if kw.get("f32") else torch.bfloat16)
        i = [0]
        def f():
            i[0] = (i[0] + 1) % 8
            ops.linear_ex(x, ws[i[0]], residual=out if kw.get("res") else None, out=out, swiglu=sw, rms_eps=kw.get("eps", 0.0), out_f32=bool(kw.get("f32")))
        res = []
        L.mc_gemm_set_option(b"rows_min_mb", 1)
        sweep = [("skinny", OLD), ("auto", 0)]
        if "--sweep" in sys.argv:
            sweep += [(f"S{S}{'k4' if k4 else ''}", (S << 24) | (k4 << 30)) for S in (1, 2, 3, 4, 6, 8) for k4 in (0, 1)]
        for label, dbg in sweep:
            L.mc_gemm_debug(dbg)
            res.append((label, timeit(f)))
        L.mc_gemm_debug(0)
        L.mc_gemm_set_option(b"rows_min_mb", 2)
        by = N * K * 2
        print(f"M={M:3d} {name:9s}: " + " ".join(f"{l} {t*1e6:5.1f}" for l, t in res), flush=True)
