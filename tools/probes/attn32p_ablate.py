"""Timing-only ablations of attn_prefill32p_kernel (wrong results): what halving the LDS fragment reads per MFMA, or removing the LDS-DMA, would buy.
debug bits: 512 the pipelined kernel; bits 10 - 13 a mask of ablations: half of the K / V fragment reads skipped, no LDS-DMA after the prologue,
no softmax arithmetic, no barriers / DMA waits."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import _lib, ops

L_ = _lib.lib()
BF = torch.bfloat16
B, H, L, S, D = 16, 32, 2793, 2816, 128
q = torch.randn(B, L, H, D, device="cuda").to(BF)
k = torch.randn(B, H, S, D, device="cuda").to(BF)
v = torch.randn(B, H, S, D, device="cuda").to(BF)
out = torch.empty(B * L, H * D, dtype=BF, device="cuda")
args = (q, k, v, out, B, H, H, L, S, D, (L * H * D, H * D, D), (H * S * D, D, S * D), (H * S * D, D, S * D), H * D, True)
flops = 4.0 * B * H * D * L * S * 0.5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {}
for rnd in range(5):
    for nm, dbg in (("mfma32", 0), ("pipelined", 512), ("half_lds_reads", 1 << 10), ("no_dma", 2 << 10), ("half_lds_reads+no_dma", 3 << 10),
                    ("no_softmax", 4 << 10), ("no_barriers_no_dma_waits", 8 << 10), ("no_dma+no_barriers", 10 << 10),
                    ("half_lds+no_dma+no_softmax", 7 << 10), ("all_four", 15 << 10)):
        L_.mc_attn_debug(dbg)
        for _ in range(2):
            ops.attn_prefill(*args)
        e0.record()
        for _ in range(4):
            ops.attn_prefill(*args)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(nm, []).append(e0.elapsed_time(e1) / 4 * 1e-3)
L_.mc_attn_debug(0)
rep = {nm: {"median_us": round(statistics.median(ts) * 1e6, 1), "tflops_as_if": round(flops / statistics.median(ts) / 1e12, 1)} for nm, ts in res.items()}
print(json.dumps(rep))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rep, open("gpurun_out/attn32p_ablate.json", "w"), indent=1)
