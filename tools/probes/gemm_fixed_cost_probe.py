"""Per-launch fixed cost of gemm_tile256_kernel: time vs K at fixed M, N (a + b K), with and without the epilogue (timing-only build)."""
import json, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from modelcompose_amd import _lib, ops
BF = torch.bfloat16
L = _lib.lib()
VAR = {"base": 4, "noepi": 4 + (7 << 3) + (2 << 12)}     # nt / quarter / whole-line store variants: profiles/r03_probes/README.md (not in the tree)
STAG = {}
def t_of(M, N, K, dbg, iters=8, rounds=5):
    w = ops.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)); x = torch.randn(M, K, device="cuda").to(BF)
    out = torch.empty(M, N, dtype=BF, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    L.mc_gemm_debug(dbg)
    for r in range(rounds):
        for _ in range(3): ops.linear(x, w, out=out)
        e0.record()
        for _ in range(iters): ops.linear(x, w, out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    L.mc_gemm_debug(0)
    return statistics.median(ts)
res = []
for (M, N) in ((4096, 4096), (44800, 4096), (44800, 12288)):      # 256, 2800, 8400 tiles (M a multiple of 256: the "lines" variant writes whole tiles)
    for K in (1024, 4096):
        row = {"M": M, "N": N, "K": K}
        for nm, dbg in VAR.items(): row[nm + "_us"] = round(t_of(M, N, K, dbg), 1)
        for nm, v in STAG.items():
            L.mc_gemm_set_option(b"stagger", v)
            row[nm + "_us"] = round(t_of(M, N, K, 4), 1)
        L.mc_gemm_set_option(b"stagger", 0)
        res.append(row); print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gemm_fixed_cost.json", "w"), indent=1)
