"""The four routed prefill GEMMs of one decoder layer at the headline workload's shapes, standalone - for rocprofv3 PMC passes
(FETCH_SIZE / WRITE_SIZE / MFMA counters) of gemm_tile256_kernel when the profiler cannot run the whole bench.py process
(ROCm 7.2's rocprofv3 --pmc aborts with SIGSEGV inside the profiler on the img+audio+video bench process; `--workload vision` is fine).

    python tools/gemm_layer_pmc.py [reps] [batch]

Rows: B (default: bench.py's per-GPU batch of the headline workload) samples x (97 text | 42 audio | 586 vision | 2066 video) tokens in 4 adapter groups (default, audio, vision,
video), exactly the row groups mc_llm_prefill hands to mc_gemm_grouped_bf16; q|k|v (N 12288, row_scale + RoPE / cache-scatter epilogue), o (N 4096, residual),
gate|up (N 22016, SwiGLU epilogue), down (K 11008, residual).  Prints the algorithmic bytes / flops per launch in launch order."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from modelcompose_amd import ops  # noqa: E402

BF = torch.bfloat16


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    import os
    if os.environ.get("MC_GEMM_DEBUG"):                      # A/B builds of the kernel under the counters (e.g. 12344 = without the next-tile L2 warm-up)
        from modelcompose_amd import _lib
        _lib.lib().mc_gemm_debug(int(os.environ["MC_GEMM_DEBUG"]))
    if os.environ.get("MC_GEMM_OPTIONS"):                    # e.g. "raster_slab=8,raster_auto=0" (tools/pmc_clock_vs_traffic.sh)
        from modelcompose_amd import _lib
        for kv in os.environ["MC_GEMM_OPTIONS"].split(","):
            k, v = kv.split("=")
            _lib.check(_lib.lib().mc_gemm_set_option(k.encode(), int(v)), "mc_gemm_set_option " + kv)
    import bench
    B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.WORKLOADS["iav"][2]
    Hd, I = 4096, 11008
    rows = [B * 97, B * 42, B * 586, B * 2066]
    gs = [0]
    for r in rows:
        gs.append(gs[-1] + r)
    M = gs[-1]
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g, device="cuda")
    W = lambda n, k: [ops.pack_weight((rnd(n, k) * k ** -0.5).to(BF)) for _ in range(4)]
    wqkv, wo, wgu, wd = W(3 * Hd, Hd), W(Hd, Hd), W(2 * I, Hd), W(Hd, I)
    x = rnd(M, Hd).to(BF)
    rs = torch.rand(M, device="cuda") + 0.5
    qkv = torch.empty(M, 3 * Hd, dtype=BF, device="cuda")
    attn = rnd(M, Hd).to(BF)
    inter = torch.empty(M, I, dtype=BF, device="cuda")
    # the q|k|v launch carries the RoPE + scatter epilogue of the prefill (mc_rope_scatter): rows of sample b, position t
    L, H, D = 97 + 42 + 586 + 2066, 32, 128
    Smax = (L + 32 + 63) // 64 * 64
    row_b = torch.cat([torch.arange(B, device="cuda").repeat_interleave(n) for n in (97, 42, 586, 2066)]).to(torch.int32)
    offs, parts = 0, []
    for n in (97, 42, 586, 2066):
        parts.append((torch.arange(n, device="cuda") + offs).repeat(B))
        offs += n
    row_t = torch.cat(parts).to(torch.int32)
    ang = torch.arange(Smax, dtype=torch.float32, device="cuda")[:, None] * (10000.0 ** (-torch.arange(64, dtype=torch.float32, device="cuda") / 64))[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    q_out = torch.empty(B * L, H * D, dtype=BF, device="cuda")
    kc = torch.empty(B, H, Smax, D, dtype=BF, device="cuda")
    vc = torch.empty(B, H, Smax, D, dtype=BF, device="cuda")
    rope = ops.rope_scatter(row_b, row_t, row_t, cos, sin, q_out, kc, vc, H, H, D, L, Smax)
    launches = []
    for _ in range(reps):
        ops.linear_grouped(x, wqkv, gs, row_scale=rs, out=qkv, rope=rope)
        ops.linear_grouped(attn, wo, gs, residual=x, out=x)
        ops.linear_grouped(x, wgu, gs, row_scale=rs, swiglu=True, out=inter)
        ops.linear_grouped(inter, wd, gs, residual=x, out=x)
        x.mul_(0.25)
    torch.cuda.synchronize()
    for name, N, K, nout, res in (("qkv", 3 * Hd, Hd, 3 * Hd, 0), ("o", Hd, Hd, Hd, 1), ("gate_up", 2 * I, Hd, I, 0), ("down", Hd, I, Hd, 1)):
        launches.append({"gemm": name, "M": M, "N": N, "K": K, "flops": 2.0 * M * N * K,
                         "algorithmic_bytes": 2.0 * (M * K + 4 * N * K + M * nout + res * M * Hd)})
    print(json.dumps({"reps": reps, "per_gpu_batch": B, "launch_order": launches}))


if __name__ == "__main__":
    main()
