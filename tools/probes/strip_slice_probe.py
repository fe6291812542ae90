"""Go / no-go for a K split of the strip kernel over workgroups (N = 4096 linears at 17..64 rows: one block-row per workgroup makes every
workgroup take in ALL of x): the main phase of a 4-slice launch is emulated by the shipped kernel on a (4 N, K / 4) weight - the same
weight bytes, 256 workgroups of R = 4 block-rows, a quarter of x per workgroup - and timed beside the real shape (graph replays, rotating
weights).  The hand-off (partials out, last arriver sums) is not in the emulation: ~3 MB of traffic and one dependent round trip."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import ops

BF = ops.BF16


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


rep = {}
for M in (16, 32, 48, 64):
    for (name, N, K) in (("o", 4096, 4096), ("o/4", 16384, 1024), ("o/2", 8192, 2048), ("down", 4096, 11008), ("down/4", 16384, 2752), ("down/2", 8192, 5504),
                         ("qkv", 12288, 4096), ("qkv/2", 24576, 2048), ("gate_up", 22016, 4096), ("gate_up/2", 44032, 2048)):
        ws = [ops.pack_weight(torch.randn(N, K, device="cuda", dtype=BF) * 0.02) for _ in range(8)]
        x = torch.randn(M, K, device="cuda", dtype=BF)
        out = torch.empty(M, N, device="cuda", dtype=BF)
        i = [0]
        def f():
            i[0] = (i[0] + 1) % 8
            ops.linear_ex(x, ws[i[0]], out=out)
        t = timeit(f)
        rep[f"M{M}_{name}"] = {"us": round(t * 1e6, 2), "TBps": round(N * K * 2 / t / 1e12, 2)}
        print(f"M={M:2d} {name:10s} N={N:5d} K={K:5d}: {t * 1e6:6.2f} us  {N * K * 2 / t / 1e12:.2f} TB/s", flush=True)
        del ws
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rep, open("gpurun_out/strip_slice_probe.json", "w"), indent=1)
