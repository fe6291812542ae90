"""The CU partition taken apart (profiles/r05_probes/cu_partition_ab.json lost at every N): on streams masked to CUs [N, 256) and [0, N)
  1. a prefill-shaped GEMM alone on 256 - N CUs            (does the mask itself cost more than the CUs it takes?)
  2. the decode attention alone on N CUs                    (tools/probes/decode_cu_rate.py)
  3. both at the same time                                  (what each keeps beside the other: fabric / L2 / HBM contention)
HIP events per stream; the GEMM loop is sized to outlast the attention loop."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import _lib, ops

BF = torch.bfloat16
L = _lib.lib()


def masked(first, n):
    if n >= 256:
        return torch.cuda.Stream()
    h = C.c_void_p()
    _lib.check(L.mc_stream_create_cu_range(first, n, C.byref(h)), "mc_stream_create_cu_range")
    return torch.cuda.ExternalStream(h.value)


def main():
    B, H, D, S, Smax = 48, 32, 128, 2809, 2880
    caches = [(torch.randn(B, H, Smax, D, device="cuda").to(BF), torch.randn(B, H, Smax, D, device="cuda").to(BF)) for _ in range(8)]
    qkv = torch.randn(B, 3 * H * D, device="cuda").to(BF)
    o1 = torch.empty(B, H * D, dtype=BF, device="cuda")
    lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
    ang = torch.outer(torch.arange(4096, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D)))
    cos, sin = ang.cos().cuda().contiguous(), ang.sin().cuda().contiguous()
    attn_bytes = 2.0 * B * H * S * D * 2
    M, N_, K_ = 44656, 12288, 4096
    w = ops.pack_weight((torch.randn(N_, K_, device="cuda") * K_ ** -0.5).to(BF))
    x = torch.randn(M, K_, device="cuda").to(BF)
    out = torch.empty(M, N_, dtype=BF, device="cuda")
    flops = 2.0 * M * N_ * K_

    def gemm_loop(st, n):
        with torch.cuda.stream(st):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.linear(x, w, out=out)
            e1.record()
        return e0, e1

    def attn_loop(st, n):
        with torch.cuda.stream(st):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                kc, vc = caches[i % 8]
                ops.attn_decode_rope(qkv, cos, sin, kc, vc, o1, lens, B, H, H, Smax, D)
            e1.record()
        return e0, e1

    res = {}
    full = torch.cuda.Stream()
    gemm_loop(full, 4)
    attn_loop(full, 4)
    torch.cuda.synchronize()
    e = gemm_loop(full, 12)
    torch.cuda.synchronize()
    res["gemm_256_alone_tflops"] = round(flops * 12 / (e[0].elapsed_time(e[1]) * 1e-3) / 1e12, 1)
    print(res, flush=True)
    for n in (16, 32, 64):
        sg, sa = masked(n, 256 - n), masked(0, n)
        gemm_loop(sg, 3)
        attn_loop(sa, 3)
        torch.cuda.synchronize()
        e = gemm_loop(sg, 12)
        torch.cuda.synchronize()
        tg = e[0].elapsed_time(e[1]) * 1e-3 / 12
        e = attn_loop(sa, 12)
        torch.cuda.synchronize()
        ta = e[0].elapsed_time(e[1]) * 1e-3 / 12
        # together: the attention loop sized to run about as long as 24 GEMMs
        na = max(8, int(24 * tg / ta))
        eg = gemm_loop(sg, 24)
        ea = attn_loop(sa, na)
        torch.cuda.synchronize()
        tg2 = eg[0].elapsed_time(eg[1]) * 1e-3 / 24
        ta2 = ea[0].elapsed_time(ea[1]) * 1e-3 / na
        r = {"gemm_alone_tflops": round(flops / tg / 1e12, 1), "gemm_alone_vs_cu_share": round(flops / tg / 1e12 / (res["gemm_256_alone_tflops"] * (256 - n) / 256), 3),
             "attn_alone_GBs": round(attn_bytes / ta / 1e9, 1), "gemm_beside_attn_tflops": round(flops / tg2 / 1e12, 1),
             "attn_beside_gemm_GBs": round(attn_bytes / ta2 / 1e9, 1), "gemm_kept": round(tg / tg2, 3), "attn_kept": round(ta / ta2, 3)}
        res[f"decode_cus_{n}"] = r
        print(n, r, flush=True)
    # no masks: the two loops on two ordinary streams (the shipped pipelined loop's situation)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    eg = gemm_loop(s1, 24)
    ea = attn_loop(s2, 200)
    torch.cuda.synchronize()
    res["unmasked_together"] = {"gemm_tflops": round(flops * 24 / (eg[0].elapsed_time(eg[1]) * 1e-3) / 1e12, 1),
                                "attn_GBs": round(attn_bytes * 200 / (ea[0].elapsed_time(ea[1]) * 1e-3) / 1e9, 1)}
    print(res["unmasked_together"], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/partition_interference.json", "w"), indent=1)


if __name__ == "__main__":
    main()
