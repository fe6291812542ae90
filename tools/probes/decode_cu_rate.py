"""What one CU draws from HBM in the decode kernels: the decode attention (B 48 x 32 heads x 128, 2809 cached keys, fused RoPE + append) and the
four decode GEMMs (48 rows against the Vicuna-7B weights) on streams restricted to N CUs (mc_stream_create_cu_range: N / 8 of every XCD).
Rotating buffers far larger than the Infinity Cache, HIP events on the masked stream.  Decides whether a CU partition of the pipelined loop
(model.decode_cus) can work: a kernel that reaches the HBM ceiling only through all 256 load paths (~22 GB/s per CU) takes 256 / N times
longer on N CUs; one that draws 70-90 GB/s per CU (the guide's multi-loader figures) fits the chain on 32 CUs."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import _lib, ops

BF = torch.bfloat16
L = _lib.lib()


def masked(n):
    if n >= 256:
        return torch.cuda.Stream()
    h = C.c_void_p()
    _lib.check(L.mc_stream_create_cu_range(0, n, C.byref(h)), "mc_stream_create_cu_range")
    return torch.cuda.ExternalStream(h.value)


def timed(stream, fn, reps):
    with torch.cuda.stream(stream):
        for i in range(3):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    B, H, D, S, Smax = 48, 32, 128, 2809, 2880
    caches = [(torch.randn(B, H, Smax, D, device="cuda").to(BF), torch.randn(B, H, Smax, D, device="cuda").to(BF)) for _ in range(8)]
    qkv = torch.randn(B, 3 * H * D, device="cuda").to(BF)
    o1 = torch.empty(B, H * D, dtype=BF, device="cuda")
    lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
    ang = torch.outer(torch.arange(4096, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D)))
    cos, sin = ang.cos().cuda().contiguous(), ang.sin().cuda().contiguous()
    attn_bytes = 2.0 * B * H * S * D * 2
    shapes = {"qkv": (12288, 4096, False), "o": (4096, 4096, False), "gate_up": (22016, 4096, True), "down": (4096, 11008, False)}
    weights = {k: [ops.pack_weight((torch.randn(n, kk, device="cuda") * kk ** -0.5).to(BF)) for _ in range(6)] for k, (n, kk, _) in shapes.items()}
    xs = {k: torch.randn(B, kk, device="cuda").to(BF) for k, (n, kk, _) in shapes.items()}
    res = {"attention": {}, **{k: {} for k in shapes}}
    dbg = int(os.environ.get("MC_ATTN_DEBUG", "0"))
    if dbg:
        L.mc_attn_debug(dbg)
    for n in (256, 128, 64, 32, 16):
        st = masked(n)

        def attn(i):
            kc, vc = caches[i % 8]
            ops.attn_decode_rope(qkv, cos, sin, kc, vc, o1, lens, B, H, H, Smax, D)
        t = timed(st, attn, 16)
        res["attention"][n] = {"us": round(t * 1e6, 1), "GBs": round(attn_bytes / t / 1e9, 1), "GBs_per_cu": round(attn_bytes / t / 1e9 / n, 1)}
        print("attention", n, res["attention"][n], flush=True)
        for k, (N_, K_, sw) in shapes.items():
            def gemm(i, k=k, sw=sw):
                ops.linear_ex(xs[k], weights[k][i % 6], swiglu=True) if sw else ops.linear(xs[k], weights[k][i % 6])
            t = timed(st, gemm, 18)
            by = 2.0 * N_ * K_
            res[k][n] = {"us": round(t * 1e6, 1), "GBs": round(by / t / 1e9, 1), "GBs_per_cu": round(by / t / 1e9 / n, 1)}
            print(k, n, res[k][n], flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/decode_cu_rate.json", "w"), indent=1)


if __name__ == "__main__":
    main()
