"""gemm_tile256_kernel TFLOP/s against M at the LLM's (N, K) shapes, plain epilogue, one process: does the headline workload's M = 133 968
(48 x 2791 tokens) run slower than the 44 656 the kernel studies use?"""
import json, os, sys, statistics
sys.path.insert(0, os.getcwd())
import torch
from modelcompose_amd import _lib, ops
BF = torch.bfloat16
def t_of(M, N, K, iters=4, rounds=4):
    w = ops.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)); x = torch.randn(M, K, device="cuda").to(BF)
    out = torch.empty(M, N, dtype=BF, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for r in range(rounds):
        for _ in range(2): ops.linear(x, w, out=out)
        e0.record()
        for _ in range(iters): ops.linear(x, w, out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e-3)
    return statistics.median(ts)
rows = []
for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)):
    row = {"N": N, "K": K}
    for M in (11264, 44800, 89600, 133888, 179200):
        row[f"M{M}"] = round(2.0 * M * N * K / t_of(M, N, K) / 1e12, 1)
    rows.append(row); print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/gemm_m_scaling.json", "w"), indent=1)
