"""A/B of the rows kernel's in-launch slab fold ("rows_fold") against the separate rows_reduce_kernel launch, ONE process, interleaved rounds:
the decode GEMM chain of the headline model (48 rows: q|k|v with folded RMS, o + residual, gate|up SwiGLU, down + residual) over 32 layers'
worth of distinct weights (13 GB: nothing stays in the Infinity Cache), eager and as a replayed graph."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import _lib, ops

L = _lib.lib()
BF = torch.bfloat16
Hd, I, M, NL = 4096, 11008, 48, 32
mk = lambda n, k: ops.pack_weight((torch.randn(n, k, device="cuda") * k ** -0.5).to(BF))
W = [(mk(3 * Hd, Hd), mk(Hd, Hd), mk(2 * I, Hd), mk(Hd, I)) for _ in range(NL)]
x0 = torch.randn(M, Hd, device="cuda").to(BF)
att = torch.randn(M, Hd, device="cuda").to(BF)
bufs = dict(q=torch.empty(M, 3 * Hd, dtype=BF, device="cuda"), h1=torch.empty(M, Hd, dtype=BF, device="cuda"),
            it=torch.empty(M, I, dtype=BF, device="cuda"), h2=torch.empty(M, Hd, dtype=BF, device="cuda"))


def step():
    x = x0
    for (wq, wo, wg, wd) in W:
        ops.linear_ex(x, wq, rms_eps=1e-5, out=bufs["q"])
        ops.linear_ex(att, wo, residual=x, out=bufs["h1"])
        ops.linear_ex(bufs["h1"], wg, rms_eps=1e-5, swiglu=True, out=bufs["it"])
        ops.linear_ex(bufs["it"], wd, residual=bufs["h1"], out=bufs["h2"])
        x = bufs["h2"]


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {}
graphs = {}
for fold in (0, 1):
    _lib.check(L.mc_gemm_set_option(b"rows_fold", fold), "rows_fold")
    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    graphs[fold] = g
for rnd in range(6):
    for fold in ((0, 1) if rnd % 2 == 0 else (1, 0)):
        _lib.check(L.mc_gemm_set_option(b"rows_fold", fold), "rows_fold")
        res.setdefault(("eager", fold), []).append(timed(step, 3))
        res.setdefault(("graph", fold), []).append(timed(graphs[fold].replay, 5))
_lib.check(L.mc_gemm_set_option(b"rows_fold", 0), "rows_fold")
by = 2.0 * (3 * Hd * Hd + Hd * Hd + 2 * I * Hd + Hd * I) * NL
rep = {f"{k}_fold{f}": {"us_per_32_layers": round(statistics.median(v), 1), "us_per_layer": round(statistics.median(v) / NL, 2),
                        "GBs": round(by / statistics.median(v) / 1e3, 1)} for (k, f), v in res.items()}
print(json.dumps(rep))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rep, open("gpurun_out/rows_fold_ab.json", "w"), indent=1)
