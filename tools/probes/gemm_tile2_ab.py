"""A/B of the two large-M tile economies in ONE process, interleaved rounds, random data (guide rules 24 / 25):
  t256     gemm_tile256_kernel: 256 x 256 x 64 tiles, 8 waves, one workgroup per CU (the shipped kernel)
  t2_v0/1/2 gemm_tile2_kernel: 256 x 128 x 32 tiles, 4 waves, TWO workgroups per CU, 3-stage LDS-DMA ring; the K-step's six DMA instructions
           in front of the fragment reads (v0), behind them (v1), or inside the MFMA stream (v2)
Shapes: the four LLM linears of the headline workload at M = 44 656 (a third of the B = 48 prefill: same tile counts per CU round), the
encoder shapes at K = 1024 (CLIP / LanguageBind towers: the 256 x 256 kernel's prologue + epilogue are 28 % of a tile there), 8192^3.
First every variant is compared bit for bit with the shipped kernel, then a race screen (the same launch 60 times), then the timing."""
import json
import os
import random
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import _lib, ops

BF = torch.bfloat16
L = _lib.lib()
VARIANTS = {"t256": (0, 0), "t2_v0": (1, 0), "t2_v1": (1, 1), "t2_v2": (1, 2)}


def select(name):
    on, var = VARIANTS[name]
    _lib.check(L.mc_gemm_set_option(b"tile2", on), "tile2")
    _lib.check(L.mc_gemm_set_option(b"tile2_variant", var), "tile2_variant")
    L.mc_gemm_debug(4)


def reset():
    L.mc_gemm_debug(0)
    L.mc_gemm_set_option(b"tile2", 0)
    L.mc_gemm_set_option(b"tile2_variant", 0)


def check():
    ok = True
    g = torch.Generator(device="cuda").manual_seed(5)
    for (M, N, K) in ((256, 256, 128), (300, 520, 192), (1000, 4096, 1024), (4096, 4096, 4096), (513, 1028, 11008), (10928, 12288, 4096), (2000, 768, 256), (700, 1024, 64), (700, 1024, 32 * 3)):
        if K % 64:
            continue
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(BF)
        x = torch.randn(M, K, device="cuda", generator=g).to(BF)
        res = torch.randn(M, N, device="cuda", generator=g).to(BF)
        bias = torch.randn(N, device="cuda", generator=g).to(BF)
        pw = ops.pack_weight(w, bias)
        outs = {}
        for nm in VARIANTS:
            select(nm)
            outs[nm] = [ops.linear(x, pw, residual=res), ops.linear(x, pw, act="quick_gelu"), ops.linear(x, pw, bias=False)]
        reset()
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for nm in VARIANTS for a, b in zip(outs["t256"], outs[nm]))
        ref = x.float() @ w.float().t()
        err = ((outs["t2_v0"][2].float() - ref).abs().max() / ref.abs().max()).item()
        print(f"check M={M} N={N} K={K}: tile2 variants bit-identical to the 256 x 256 kernel: {same}; rel err vs fp32 {err:.2e}", flush=True)
        ok &= same and err < 1e-2
    w = (torch.randn(4096, 4096, device="cuda", generator=g) * 4096 ** -0.5).to(BF)
    x = torch.randn(8192, 4096, device="cuda", generator=g).to(BF)
    w2 = (torch.randn(1024, 1024, device="cuda", generator=g) * 1024 ** -0.5).to(BF)
    x2 = torch.randn(30000, 1024, device="cuda", generator=g).to(BF)
    bad = 0
    for (xx, ww) in ((x, w), (x2, w2)):
        pw = ops.pack_weight(ww)
        select("t256")
        ref = ops.linear(xx, pw)
        for nm in list(VARIANTS)[1:]:
            select(nm)
            for _ in range(60):
                bad += int(not torch.equal(ops.linear(xx, pw), ref))
    reset()
    print(f"race screen: {bad} of 360 tile2 launches differ from the 256 x 256 kernel's output", flush=True)
    return ok and bad == 0


def bench(shapes, variants, rounds=6, iters=6):
    res, bufs = {}, {}
    for (M, N, K) in shapes:
        w = ops.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).to(BF))
        x = torch.randn(M, K, device="cuda").to(BF)
        out = torch.empty(M, N, dtype=BF, device="cuda")
        bufs[(M, N, K)] = (w, x, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w, x, out = bufs[shapes[0]]
    for _ in range(30):
        ops.linear(x, w, out=out)
    torch.cuda.synchronize()
    for r in range(rounds):
        for shp in shapes:
            w, x, out = bufs[shp]
            order = list(variants)
            random.Random(1000 * r + len(res)).shuffle(order)
            for nm in order:
                select(nm)
                for _ in range(3):
                    ops.linear(x, w, out=out)
                e0.record()
                for _ in range(iters):
                    ops.linear(x, w, out=out)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault((shp, nm), []).append(e0.elapsed_time(e1) / iters * 1e-3)
    reset()
    table = []
    for shp in shapes:
        M, N, K = shp
        row = {"M": M, "N": N, "K": K}
        for nm in variants:
            ts = res[(shp, nm)]
            row[nm] = {"median_tflops": round(2.0 * M * N * K / statistics.median(ts) / 1e12, 1), "best_tflops": round(2.0 * M * N * K / min(ts) / 1e12, 1),
                       "median_us": round(statistics.median(ts) * 1e6, 1)}
        table.append(row)
        print(json.dumps(row), flush=True)
    return table


if __name__ == "__main__":
    ok = check()
    shapes = [(8192, 8192, 8192), (44656, 12288, 4096), (44656, 4096, 4096), (44656, 22016, 4096), (44656, 4096, 11008),
              (27696, 3072, 1024), (27696, 1024, 1024), (27696, 4096, 1024), (27696, 1024, 4096),          # CLIP-L tower, B = 48: 577 tokens per image
              (98688, 3072, 1024), (98688, 1024, 1024), (98688, 4096, 1024), (98688, 1024, 4096),          # LanguageBind-Video, B = 48: 8 x 257 tokens
              (24576, 2304, 768), (24576, 3072, 768), (24576, 768, 3072)]                                   # BEATs, B = 48: 512 tokens
    t = bench(shapes, list(VARIANTS))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"bit_identical_and_race_free": ok, "table": t}, open("gpurun_out/gemm_tile2_ab.json", "w"), indent=1)
    print("tile2 bit-identical and race-free:", ok)
