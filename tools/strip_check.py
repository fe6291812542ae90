"""gemm_strip_kernel (every M <= 64 linear) and the chunked decode attention on a GPU: correctness against fp32 torch, bitwise batch
invariance (row m of an M-row launch == the same row launched alone / in a smaller batch), and timings against the round-5 kernels
(numbers of record: profiles/r06_probes/strip_check_vs_round5.log) as graph replays over rotating weights.  Writes gpurun_out/strip_check.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import _lib, ops

L = _lib.lib()
BF = ops.BF16
rep = {"storage": _lib.storage_name(), "mismatch": 0, "invariance_failures": 0, "timing_us": {}}
SHAPES = ((4096, 4096, "o"), (4096, 11008, "down"), (12288, 4096, "qkv"), (22016, 4096, "gate_up"), (32000, 4096, "head"), (1000, 256, "small"),
          (96, 64, "tiny"), (4096, 4160, "odd_k"))


def variants(N):
    v = [("plain", {}), ("res", dict(residual=True)), ("rms", dict(rms_eps=1e-5)), ("f32", dict(out_f32=True))]
    if N % 32 == 0:
        v.append(("swiglu+rms", dict(swiglu=True, rms_eps=1e-5)))
    return v


def run(x, w, res, kw, strip=1):
    """strip = 0: the tile family (the other summation order) as the second opinion"""
    kw = dict(kw)
    if kw.pop("residual", False):
        kw["residual"] = res[: x.shape[0]]
    if not strip:
        eps = kw.pop("rms_eps", 0.0)
        if eps:
            kw["row_scale"] = torch.rsqrt(x.float().pow(2).mean(-1) * (x.shape[1] / w.K) + eps).contiguous()
        kw["family"] = "tile"
    return ops.linear_ex(x, w, **kw)


torch.manual_seed(0)
ATTN_ONLY = "--attn-only" in sys.argv
for (N, K, name) in (() if ATTN_ONLY else SHAPES):
    wd = torch.randn(N, K, device="cuda", dtype=BF) * 0.02
    w = ops.pack_weight(wd)
    Kp = w.Kp
    x64 = torch.zeros(64, Kp, device="cuda", dtype=BF)
    x64[:, :K] = torch.randn(64, K, device="cuda", dtype=BF)
    res = torch.randn(64, N, device="cuda", dtype=BF)
    for vn, kw in variants(N):
        full = run(x64, w, res, kw)
        # truth (plain only)
        if vn == "plain":
            truth = x64[:, :K].float() @ wd.float().t()
            e = (full.float() - truth).abs().max().item() / truth.abs().max().item()
            old = run(x64, w, res, kw, strip=0)
            eo = (old.float() - truth).abs().max().item() / truth.abs().max().item()
            print(f"{name:8s} N={N} K={K}: rel err vs fp32  strip {e:.2e}  tile family {eo:.2e}", flush=True)
            if not e < 8e-3:
                rep["mismatch"] += 1
        else:
            old = run(x64, w, res, kw, strip=0)
            e = (full.float() - old.float()).abs().max().item() / max(old.float().abs().max().item(), 1e-6)
            if not e < 1.5e-2:
                rep["mismatch"] += 1
                print(f"MISMATCH {name} {vn}: {e:.3e}")
        # bitwise invariance: rows of smaller launches
        for M in (1, 2, 8, 15, 16, 17, 31, 33, 47, 48, 63):
            y = run(x64[:M].contiguous(), w, res, kw)
            if not torch.equal(y, full[:M]):
                rep["invariance_failures"] += 1
                print(f"INVARIANCE {name} {vn} M={M}: max diff {(y.float() - full[:M].float()).abs().max().item():.3e}")
        # the strip family above 64 rows: slices
        if vn in ("plain", "rms"):
            x100 = torch.cat([x64, x64[:36]], 0).contiguous()
            y = ops.linear_ex(x100, w, family="strip", **kw)
            if not (torch.equal(y[:64], full) and torch.equal(y[64:], full[:36])):
                rep["invariance_failures"] += 1
                print(f"INVARIANCE {name} {vn} M=100 slices")
        # padded row stride
        xp = torch.zeros(64, Kp + 64, device="cuda", dtype=BF)
        xp[:, :Kp] = x64
        y = run(xp[:, :Kp], w, res, kw)
        if not torch.equal(y, full):
            rep["invariance_failures"] += 1
            print(f"INVARIANCE {name} {vn} padded stride")
print("gemm mismatches:", rep["mismatch"], "invariance failures:", rep["invariance_failures"], flush=True)

# ---- decode attention: reference, invariance in B / nsplit / Smax
def attn_ref(q, kc, vc, lens, scale):
    B, H, D = q.shape
    out = torch.zeros(B, H, D, dtype=torch.float32, device=q.device)
    for b in range(B):
        n = int(lens[b])
        s = torch.einsum("hd,hsd->hs", q[b].float(), kc[b, :, :n].float()) * scale
        p = torch.softmax(s, -1)
        out[b] = torch.einsum("hs,hsd->hd", p, vc[b, :, :n].float())
    return out


rep["attn_failures"] = 0
if True:
    for D in (128, 64):
        B, H, S = 6, 8, 700
        q = torch.randn(B, H, D, device="cuda", dtype=BF)
        kc = torch.randn(B, H, S, D, device="cuda", dtype=BF)
        vc = torch.randn(B, H, S, D, device="cuda", dtype=BF)
        lens = torch.tensor([1, 127, 128, 129, 400, 700], dtype=torch.int32, device="cuda")
        o = torch.empty(B, H * D, device="cuda", dtype=BF)
        st = (H * S * D, S * D, D)
        ops.attn_decode(q, kc, vc, o, B, H, H, S, D, (H * D, D), (H * S * D, D, S * D), (H * S * D, D, S * D), H * D, nsplit=1, kv_lens=lens)
        ref = attn_ref(q, kc, vc, lens.tolist(), 1.0 / D ** 0.5).reshape(B, H * D)
        e = (o.float() - ref).abs().max().item()
        print(f"attn_decode D={D}: max err vs fp32 {e:.3e}")
        if not e < 3e-2:
            rep["attn_failures"] += 1
        for ns in (2, 3, 8):
            o2 = torch.empty_like(o)
            ops.attn_decode(q, kc, vc, o2, B, H, H, S, D, (H * D, D), (H * S * D, D, S * D), (H * S * D, D, S * D), H * D, nsplit=ns, kv_lens=lens)
            if not torch.equal(o, o2):
                rep["attn_failures"] += 1
                print(f"ATTN nsplit={ns} differs: {(o.float() - o2.float()).abs().max().item():.3e}")
        # one sequence alone, in a cache with another Smax
        for b in (1, 3, 5):
            S2 = 1024
            k1 = torch.zeros(1, H, S2, D, device="cuda", dtype=BF); v1 = torch.zeros_like(k1)
            k1[0, :, :S] = kc[b]; v1[0, :, :S] = vc[b]
            o1 = torch.empty(1, H * D, device="cuda", dtype=BF)
            ops.attn_decode(q[b:b + 1].contiguous(), k1, v1, o1, 1, H, H, S2, D, (H * D, D), (H * S2 * D, D, S2 * D), (H * S2 * D, D, S2 * D), H * D, nsplit=4,
                            kv_lens=lens[b:b + 1].contiguous())
            if not torch.equal(o1[0], o[b]):
                rep["attn_failures"] += 1
                print(f"ATTN row {b} alone differs: {(o1[0].float() - o[b].float()).abs().max().item():.3e}")
    print("attention failures:", rep["attn_failures"], flush=True)

if "--time" in sys.argv:
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

    for M in (() if ATTN_ONLY else (1, 8, 16, 32, 48, 64)):
        for (N, K, name, kw) in ((4096, 4096, "o_proj", dict(res=True)), (4096, 11008, "down_proj", dict(res=True)), (12288, 4096, "qkv", dict(eps=1e-5)),
                                 (22016, 4096, "gate|up", dict(eps=1e-5, sw=True)), (32000, 4096, "lm_head", dict(f32=True))):
            ws = [ops.pack_weight(torch.randn(N, K, device="cuda", dtype=BF) * 0.02) for _ in range(8)]
            for pad in (0, 64):
                xb = torch.randn(M, K + pad, device="cuda", dtype=BF)
                x = xb[:, :K]
                sw = kw.get("sw", False)
                out = torch.randn(M, N // 2 if sw else N, device="cuda", dtype=torch.float32 if kw.get("f32") else BF)
                i = [0]
                def f():
                    i[0] = (i[0] + 1) % 8
                    ops.linear_ex(x, ws[i[0]], residual=out if kw.get("res") else None, out=out, swiglu=sw, rms_eps=kw.get("eps", 0.0), out_f32=bool(kw.get("f32")))
                row = {}
                t = timeit(f)
                row["strip"] = round(t * 1e6, 2)
                row["strip_TBps"] = round(N * K * 2 / t / 1e12, 2)
                rep["timing_us"][f"M{M}_{name}_pad{pad}"] = row
                print(f"M={M:2d} {name:9s} pad={pad:2d}: {row}", flush=True)
            del ws
    # decode attention at the benchmarked shape
    for (B, kv) in ((48, 2810), (48, 2049), (8, 2810), (1, 2810)):
        H, D, S = 32, 128, 2880
        qkv = torch.randn(B, 3 * H * D, device="cuda", dtype=BF)
        kc = torch.randn(B, H, S, D, device="cuda", dtype=BF); vc = torch.randn(B, H, S, D, device="cuda", dtype=BF)
        cos = torch.randn(4096, D // 2, device="cuda"); sin = torch.randn(4096, D // 2, device="cuda")
        lens = torch.full((B,), kv, dtype=torch.int32, device="cuda")
        o = torch.empty(B, H * D, device="cuda", dtype=BF)
        ws = ops.decode_workspace(B, H, D, S, "cuda")
        for ns in ((1,) if B >= 16 else (1, 2, 4, 6, 8, 16)):
            t = timeit(lambda: ops.attn_decode_rope(qkv, cos, sin, kc, vc, o, lens, B, H, H, S, D, nsplit=ns, workspace=ws), n=100)
            by = 2 * B * H * kv * D * 2
            rep["timing_us"][f"attn_B{B}_ns{ns}"] = {"us": round(t * 1e6, 2), "TBps": round(by / t / 1e12, 2)}
            print(f"attn decode B={B} nsplit={ns}: {t * 1e6:.1f} us  {by / t / 1e12:.2f} TB/s", flush=True)

os.makedirs("gpurun_out", exist_ok=True)
tag = os.environ.get("STRIP_CHECK_TAG", "")
json.dump(rep, open(f"gpurun_out/strip_check{tag}.json", "w"), indent=1)
sys.exit(1 if (rep["mismatch"] or rep["invariance_failures"] or rep["attn_failures"]) else 0)
