"""The four routed layer GEMMs of the headline workload exactly as mc_llm_prefill launches them (4 adapter groups, row_scale, RoPE / cache
scatter, residual, SwiGLU) against the SAME shapes with one group and the plain epilogue, in one process: what the model's epilogues and
row groups cost gemm_tile256_kernel."""
import json, os, statistics, sys
sys.path.insert(0, os.getcwd())
import torch
import bench
from modelcompose_amd import _lib, ops
L_ = _lib.lib()
BF = torch.bfloat16
B = bench.WORKLOADS["iav"][2]
Hd, I = 4096, 11008
rows = [B * 97, B * 42, B * 586, B * 2066]
gs = [0]
for r in rows: gs.append(gs[-1] + r)
M = gs[-1]
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device="cuda")
W = lambda n, k: [ops.pack_weight((rnd(n, k) * k ** -0.5).to(BF)) for _ in range(4)]
wqkv, wo, wgu, wd = W(3 * Hd, Hd), W(Hd, Hd), W(2 * I, Hd), W(Hd, I)
x = rnd(M, Hd).to(BF); rs = torch.rand(M, device="cuda") + 0.5
qkv = torch.empty(M, 3 * Hd, dtype=BF, device="cuda"); attn = rnd(M, Hd).to(BF); inter = torch.empty(M, I, dtype=BF, device="cuda")
gu_plain = torch.empty(M, 2 * I, dtype=BF, device="cuda"); out_h = torch.empty(M, Hd, dtype=BF, device="cuda"); inter_in = rnd(M, I).to(BF)
L, H, D = 97 + 42 + 586 + 2066, 32, 128
Smax = (L + 32 + 63) // 64 * 64
row_b = torch.cat([torch.arange(B, device="cuda").repeat_interleave(n) for n in (97, 42, 586, 2066)]).to(torch.int32)
offs, parts = 0, []
for n in (97, 42, 586, 2066):
    parts.append((torch.arange(n, device="cuda") + offs).repeat(B)); offs += n
row_t = torch.cat(parts).to(torch.int32)
ang = torch.arange(Smax, dtype=torch.float32, device="cuda")[:, None] * (10000.0 ** (-torch.arange(64, dtype=torch.float32, device="cuda") / 64))[None]
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
q_out = torch.empty(B * L, H * D, dtype=BF, device="cuda"); kc = torch.empty(B, H, Smax, D, dtype=BF, device="cuda"); vc = torch.empty(B, H, Smax, D, dtype=BF, device="cuda")
rope = ops.rope_scatter(row_b, row_t, row_t, cos, sin, q_out, kc, vc, H, H, D, L, Smax)
one = [0, M]
cases = {
    "qkv": (3 * Hd, Hd, lambda: ops.linear_grouped(x, wqkv, gs, row_scale=rs, out=qkv, rope=rope), lambda: ops.linear_grouped(x, wqkv[:1], one, out=qkv),
            lambda: ops.linear_grouped(x, wqkv, gs, out=qkv)),
    "o": (Hd, Hd, lambda: ops.linear_grouped(attn, wo, gs, residual=x, out=out_h), lambda: ops.linear_grouped(attn, wo[:1], one, out=out_h), lambda: ops.linear_grouped(attn, wo, gs, out=out_h)),
    "gate_up": (2 * I, Hd, lambda: ops.linear_grouped(x, wgu, gs, row_scale=rs, swiglu=True, out=inter), lambda: ops.linear_grouped(x, wgu[:1], one, out=gu_plain),
                lambda: ops.linear_grouped(x, wgu, gs, out=gu_plain)),
    "down": (Hd, I, lambda: ops.linear_grouped(inter_in, wd, gs, residual=x, out=out_h), lambda: ops.linear_grouped(inter_in, wd[:1], one, out=out_h),
             lambda: ops.linear_grouped(inter_in, wd, gs, out=out_h)),
}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def t_of(f, iters=4):
    for _ in range(2): f()
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
res = {}
for rnd_ in range(4):
    for name, (N, K, real, plain, grouped) in cases.items():
        for nm, f in (("model", real), ("plain", plain), ("groups_only", grouped), ("model_res8", real)):
            L_.mc_gemm_debug((7 << 3) + (4 << 12) if nm == "model_res8" else 0)      # A/B build: residual rows in 8-byte loads (round 2)
            res.setdefault((name, nm), []).append(t_of(f))
            L_.mc_gemm_debug(0)
out = []
for name, (N, K, *_r) in cases.items():
    row = {"gemm": name, "M": M, "N": N, "K": K}
    for nm in ("model", "plain", "groups_only", "model_res8"):
        row[nm + "_tflops"] = round(2.0 * M * N * K / statistics.median(res[(name, nm)]) / 1e12, 1)
    out.append(row); print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_layer_vs_plain.json", "w"), indent=1)
