import json, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from modelcompose_amd import _lib, ops
BF = torch.bfloat16
L = _lib.lib()
def t_of(M, N, K, dbg, act=None, iters=10, rounds=5):
    w = ops.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).to(BF), torch.randn(N, device="cuda").to(BF)); x = torch.randn(M, K, device="cuda").to(BF)
    out = torch.empty(M, N, dtype=BF, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    L.mc_gemm_debug(dbg)
    for r in range(rounds):
        for _ in range(3): ops.linear(x, w, out=out, act=act)
        e0.record()
        for _ in range(iters): ops.linear(x, w, out=out, act=act)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    L.mc_gemm_debug(0)
    return statistics.median(ts)
# encoder shapes of the headline workload: video tower M = 48*8*257 = 98688, CLIP M = 48*577 = 27696
for (M, N, K, act) in ((98688, 3072, 1024, None), (98688, 1024, 1024, None), (98688, 4096, 1024, "quick_gelu"), (98688, 1024, 4096, None), (27696, 3072, 1024, None), (27696, 4096, 1024, "quick_gelu")):
    row = {"M": M, "N": N, "K": K}
    for nm, dbg in (("auto", 0), ("t256", 4), ("t192", 1024), ("t128", 2)):
        t = t_of(M, N, K, dbg, act)
        row[nm] = round(2.0 * M * N * K / t / 1e6, 1)
    print(json.dumps(row), flush=True)
