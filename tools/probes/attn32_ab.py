"""A/B of the two prefill attention kernels of the LLM shape (head_dim 128, whole key tiles, no per-key mask) in ONE process, interleaved
rounds, gaussian data (guide rules 24 / 25): `mfma16` = attn_prefill_kernel<128, .., 4 waves x 2 query blocks> on v_mfma_f32_16x16x32
(mc_attn_debug bit 7), `mfma32` = attn_prefill32_kernel on v_mfma_f32_32x32x16 (round 5).  FLOPs counted causal: 4 B H D L S / 2.
Needs the probes build: make -C modelcompose_amd/csrc probes; MC_PROBES_LIB=1.  (The software-pipelined third kernel of round 5 lost this A/B -
profiles/r05_probes/attn32_ab*.json - and was removed in round 6.)"""
import json
import os
import random
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import _lib, ops

L_ = _lib.lib()
BF = torch.bfloat16


def run(shapes, rounds=6, iters=4):
    res = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for (B, H, L, S, causal) in shapes:
        D = 128
        q = torch.randn(B, L, H, D, device="cuda").to(BF)
        k = torch.randn(B, H, S, D, device="cuda").to(BF)
        v = torch.randn(B, H, S, D, device="cuda").to(BF)
        out = torch.empty(B * L, H * D, dtype=BF, device="cuda")
        args = (q, k, v, out, B, H, H, L, S, D, (L * H * D, H * D, D), (H * S * D, D, S * D), (H * S * D, D, S * D), H * D, causal)
        outs = {}
        for r in range(rounds):
            order = ["mfma16", "mfma32"]
            random.Random(r).shuffle(order)
            for nm in order:
                L_.mc_attn_debug({"mfma16": 128, "mfma32": 256}[nm])
                for _ in range(2):
                    ops.attn_prefill(*args)
                e0.record()
                for _ in range(iters):
                    ops.attn_prefill(*args)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(((B, H, L, S, causal), nm), []).append(e0.elapsed_time(e1) / iters * 1e-3)
                outs[nm] = out.clone()
        L_.mc_attn_debug(0)
        d = (outs["mfma16"].float() - outs["mfma32"].float()).abs().max().item()
        flops = 4.0 * B * H * D * L * S * (0.5 if causal else 1.0)
        row = {"B": B, "H": H, "L": L, "S": S, "causal": causal, "max_abs_diff_between_kernels": d}
        for nm in ("mfma16", "mfma32"):
            ts = res[((B, H, L, S, causal), nm)]
            row[nm] = {"median_tflops": round(flops / statistics.median(ts) / 1e12, 1), "best_tflops": round(flops / min(ts) / 1e12, 1),
                       "median_us": round(statistics.median(ts) * 1e6, 1)}
        print(json.dumps(row), flush=True)
        res[(B, H, L, S, causal)] = row
        del q, k, v, out
    return [v for kk, v in res.items() if not isinstance(kk[-1], str)]


if __name__ == "__main__":
    shapes = [(16, 32, 2793, 2816, True), (48, 32, 2793, 2816, True), (16, 32, 683, 704, True), (48, 32, 683, 704, True), (16, 32, 2048, 2048, True),
              (8, 32, 3337, 3392, True), (16, 32, 2304, 2304, False)]
    t = run(shapes)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(t, open("gpurun_out/attn32_ab.json", "w"), indent=1)
