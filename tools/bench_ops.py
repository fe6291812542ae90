"""Micro-benchmarks of the hot kernels at Vicuna-7B shapes (run on the GPU box)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import ops

BF = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def gemm_case(M, N, K, rot=4):
    # rotate over several weight copies so decode shapes are not served from the 256 MiB Infinity Cache
    ws = [ops.pack_weight(torch.randn(N, K, device="cuda").to(BF) * K ** -0.5) for _ in range(rot)]
    x = torch.randn(M, K, device="cuda").to(BF)
    out = torch.empty(M, N, dtype=BF, device="cuda")
    i = [0]
    def f():
        ops.linear(x, ws[i[0] % rot], out=out)
        i[0] += 1
    t = timeit(f)
    fl = 2.0 * M * N * K
    by = N * K * 2 + M * K * 2 + M * N * 2
    print(f"gemm M={M:6d} N={N:6d} K={K:6d}: {t*1e6:9.1f} us  {fl/t/1e12:8.1f} TFLOP/s  {by/t/1e12:6.2f} TB/s")


def main():
    which = sys.argv[1:] or ["prefill", "decode", "attn"]
    if "dbg" in which:
        import ctypes
        from modelcompose_amd import _lib
        L = _lib.lib()
        for d in (0, 1):
            L.mc_gemm_debug(d)
            print("gemm debug mode", d)
            for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)):
                gemm_case(9376, N, K, rot=1)
        L.mc_gemm_debug(0)
    if "g256" in which:
        from modelcompose_amd import _lib
        L = _lib.lib()
        # correctness of the 256x256 kernel (edge tiles, small K, epilogue variants) against an fp32 matmul of the bf16 values
        for (M, N, K) in ((256, 256, 128), (300, 520, 192), (1000, 4096, 1024), (10928, 4096, 4096), (513, 1028, 11008)):
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(BF)
            x = torch.randn(M, K, device="cuda").to(BF)
            res = torch.randn(M, N, device="cuda").to(BF)
            bias = torch.randn(N, device="cuda").to(BF)
            pw = ops.pack_weight(w, bias)
            ref = x.float() @ w.float().t() + bias.float() + res.float()
            outs = {}
            for d in (2, 4):
                L.mc_gemm_debug(d)
                outs[d] = ops.linear(x, pw, residual=res)
            L.mc_gemm_debug(0)
            torch.cuda.synchronize()
            e2 = (outs[2].float() - ref).abs().max().item() / ref.abs().max().item()
            e4 = (outs[4].float() - ref).abs().max().item() / ref.abs().max().item()
            same = torch.equal(outs[2], outs[4])
            print(f"check M={M} N={N} K={K}: rel err 128-kernel {e2:.2e}  256-kernel {e4:.2e}  bit-identical={same}")
        for rep in range(2):
            for d in (2, 4):
                L.mc_gemm_debug(d)
                print("gemm debug mode", d, "(2 = 128x128 kernel, 4 = 256x256 kernel)")
                for M in (10928,):
                    for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)):
                        gemm_case(M, N, K, rot=1)
                gemm_case(9232, 4096, 1024, rot=1); gemm_case(9232, 1024, 4096, rot=1); gemm_case(8192, 8192, 8192, rot=1)
        L.mc_gemm_debug(0)
    if "var" in which:
        from modelcompose_amd import _lib
        L = _lib.lib()
        w = (torch.randn(4096, 4096, device="cuda") * 4096 ** -0.5).to(BF)
        x = torch.randn(3000, 4096, device="cuda").to(BF)
        pw = ops.pack_weight(w)
        L.mc_gemm_debug(2); ref = ops.linear(x, pw)
        for d, nm in ((4, "DMA in load half"), (4 + 192, "balanced reads 8,4,8,4")):
            L.mc_gemm_debug(d)
            print(nm, "bit-identical to 128 kernel:", torch.equal(ops.linear(x, pw), ref))
        for rep in range(3):
            for d, nm in ((4, "DMA in load half"), (4 + 192, "balanced reads 8,4,8,4")):
                L.mc_gemm_debug(d)
                print("256x256 kernel variant:", nm)
                gemm_case(8192, 8192, 8192, rot=1)
                gemm_case(10928, 4096, 4096, rot=1)
                gemm_case(10928, 22016, 4096, rot=1)
        L.mc_gemm_debug(0)
    if "mid" in which:
        from modelcompose_amd import _lib
        L = _lib.lib()
        for rep in range(2):
            for d in (2, 4):
                L.mc_gemm_debug(d)
                print("gemm debug mode", d, "(2 = 128x128 kernel, 4 = 256x256 kernel)")
                for M in (2732, 1552, 4096):
                    for (N, K) in ((4096, 4096), (4096, 11008), (4096, 12288), (4096, 22016), (1024, 4096), (768, 4096)):
                        gemm_case(M, N, K, rot=1)
        L.mc_gemm_debug(0)
    if "abl" in which:
        from modelcompose_amd import _lib
        L = _lib.lib()
        for rep in range(2):
            for d, nm in ((4, "full"), (4 + 8, "no LDS-DMA"), (4 + 16, "no ds_read"), (4 + 24, "MFMA + barriers only"), (4 + 32, "DMA always from K-tiles 0/1 (cache-hot)")):
                L.mc_gemm_debug(d)
                print("256x256 kernel ablation:", nm)
                gemm_case(8192, 8192, 8192, rot=1)
                gemm_case(10928, 4096, 4096, rot=1)
        L.mc_gemm_debug(0)
    if "epi" in which:
        # cost of the epilogue variants of the 256x256 kernel at the headline's prefill shapes: plain store, + residual (o / down), + row_scale (q|k|v)
        M = 44656
        for (N, K) in ((4096, 4096), (4096, 11008), (12288, 4096)):
            w = ops.pack_weight((torch.randn(N, K, device="cuda") * K ** -0.5).to(BF))
            x = torch.randn(M, K, device="cuda").to(BF)
            res = torch.randn(M, N, device="cuda").to(BF)
            rs = torch.rand(M, device="cuda") + 0.5
            out = torch.empty(M, N, dtype=BF, device="cuda")
            fl = 2.0 * M * N * K
            for rep in range(2):
                for nm, f in (("plain", lambda: ops.linear(x, w, out=out)), ("residual", lambda: ops.linear(x, w, residual=res, out=out)),
                              ("residual in place", lambda: ops.linear(x, w, residual=out, out=out)),
                              ("row_scale", lambda: ops.linear_ex(x, w, row_scale=rs, out=out))):
                    t = timeit(f)
                    print(f"epilogue {nm:18s} M={M} N={N} K={K}: {t*1e6:8.1f} us {fl/t/1e12:7.1f} TFLOP/s")
    if "raster" in which:
        # A/B of the XCD tile order of the 256x256 kernel at the metric workload's prefill shapes (M = 16 x 2793 rows): debug bit 16 = every
        # XCD owns a contiguous eighth of the tile order, bit 17 = 32-tile blocks dealt round-robin (all XCDs on the same m-group)
        from modelcompose_amd import _lib
        L = _lib.lib()
        Ms = [int(a) for a in which if a.isdigit()] or [44688]
        for rep in range(2):
            for d, nm in ((65536, "raster 0: contiguous chunk per XCD"), (131072, "raster 1: blocks round-robin over XCDs")):
                L.mc_gemm_debug(d)
                print(nm)
                for M in Ms:
                    for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)):
                        gemm_case(M, N, K, rot=1)
        L.mc_gemm_debug(0)
    if "blas" in which:
        # context only (not on the product path): the vendor library GEMM torch.matmul dispatches to (hipBLASLt / rocBLAS) on the same
        # shapes and random data, next to the hand-written 256x256 kernel
        pass
        for (M, N, K) in ((10928, 12288, 4096), (10928, 4096, 4096), (10928, 22016, 4096), (10928, 4096, 11008), (8192, 8192, 8192), (2728, 4096, 4096)):
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
            pw = ops.pack_weight(w)
            res = {}
            for name, fn in (("vendor (torch.matmul)", lambda: torch.matmul(x, w.t())), ("libmc_hip", lambda: ops.linear(x, pw))):
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                best = 1e9
                for rep in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 20)
                res[name] = best
            fl = 2.0 * M * N * K
            print(f"gemm M={M} N={N} K={K}: " + "   ".join(f"{k} {v*1e3:7.1f} us {fl/v/1e9:7.0f} TFLOP/s" for k, v in res.items()))
    if "clip" in which:
        # encoder-tower shapes (CLIP-L/336: 16 x 577 rows; LanguageBind video: 32 frames x 257 rows), kernel variants on one device
        from modelcompose_amd import _lib
        L = _lib.lib()
        for rep in range(2):
            for d, nm in ((2, "128x128 kernel"), (4 + 2048, "256-row kernel, 256-column tiles"), (4 + 1024, "256-row kernel, 192-column tiles"), (0, "library choice")):
                L.mc_gemm_debug(d)
                print(nm)
                for (M, N, K) in ((9232, 3072, 1024), (9232, 1024, 1024), (9232, 4096, 1024), (9232, 1024, 4096), (8224, 3072, 1024), (8224, 1024, 1024), (8224, 4096, 1024), (8224, 1024, 4096)):
                    gemm_case(M, N, K, rot=1)
        L.mc_gemm_debug(0)
    if "lb" in which:
        from modelcompose_amd import _lib
        L = _lib.lib()
        for rep in range(3):
            for d, nm in ((4 + 2048, "launch_bounds(512,2): 128 VGPRs + AGPRs"), (4 + 2048 + 48, "launch_bounds(512,1): 224 VGPRs")):
                L.mc_gemm_debug(d)
                print(nm)
                for (M, N, K) in ((10928, 12288, 4096), (10928, 4096, 4096), (10928, 22016, 4096), (10928, 4096, 11008), (8192, 8192, 8192)):
                    gemm_case(M, N, K, rot=1)
        L.mc_gemm_debug(0)
    if "t192" in which:
        # 256-column vs 192-column tiles of the large-M kernel on the under-filled shapes of the finetune step / small-batch prefill
        from modelcompose_amd import _lib
        L = _lib.lib()
        for rep in range(2):
            for d, nm in ((4 + 2048, "256-column tiles"), (4 + 1024, "192-column tiles"), (0, "library choice")):
                L.mc_gemm_debug(d)
                print(nm)
                for (M, N, K) in ((2728, 4096, 4096), (2728, 4096, 11008), (2728, 4096, 22016), (2728, 4096, 12288), (2728, 11008, 4096), (9232, 1024, 4096), (2792, 4096, 4096)):
                    gemm_case(M, N, K, rot=1)
        L.mc_gemm_debug(0)
    if "clock" in which:
        # shader clock the chip holds inside the 256x256 kernel's main loop after >= 2 s of back-to-back launches on random data
        import ctypes as C
        from modelcompose_amd import _lib
        L = _lib.lib()
        for (M, N, K) in ((10928, 4096, 4096), (10928, 22016, 4096), (8192, 8192, 8192)):
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            w = ops.pack_weight(torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02)
            L.mc_gemm_debug(4 + 40)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 2.5:
                for _ in range(20):
                    ops.linear(x, w)
                torch.cuda.synchronize()
                n += 20
            dt = (time.perf_counter() - t0) / n
            ghz = C.c_double()
            _lib.check(L.mc_gemm_clock_read(min(4096, ((M + 255) // 256) * ((N + 255) // 256)), C.byref(ghz)), "mc_gemm_clock_read")
            L.mc_gemm_debug(0)
            tf = 2.0 * M * N * K / dt / 1e12
            print(f"clock M={M} N={N} K={K}: {ghz.value:.3f} GHz in the main loop; {tf:.0f} TFLOP/s (host-timed incl. launch); "
                  f"MFMA peak at that clock {2.5e3 * ghz.value / 2.4:.0f} TFLOP/s -> {tf / (2.5e3 * ghz.value / 2.4):.2f} of it")
    if "ties" in which:
        from modelcompose_amd import compose
        n, d = 3, 327_155_712                  # the rank-128 default adapter of Vicuna-7B: 0.33 G elements per checkpoint
        for dt in (torch.bfloat16, torch.float32):
            flat = (torch.randn(n, d, device="cuda", dtype=torch.float32) * 0.02).to(dt)
            compose.ties_merge_vectors(flat, 20, "mean")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            compose.ties_merge_vectors(flat, 20, "mean")
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            by = flat.numel() * flat.element_size()
            # 3 histogram passes + sign election + merge read the task vectors 5 times; sign [d] int8 written + read, out [d] written
            alg = 5 * by + 2 * d + d * flat.element_size()
            print(f"ties-mean n={n} d={d} {dt}: {t*1e3:8.2f} ms  {alg/t/1e12:5.2f} TB/s over {alg/1e9:.1f} GB (5 streaming passes)")
            del flat
    if "prefill" in which:
        for M in (1536, 9376, 10912):
            for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)):
                gemm_case(M, N, K, rot=1)
        gemm_case(9232, 1024, 1024, rot=1); gemm_case(9232, 4096, 1024, rot=1); gemm_case(9232, 1024, 4096, rot=1)
    if "decode" in which:
        for M in (1, 16, 32, 64):
            for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32000, 4096)):
                gemm_case(M, N, K, rot=6)
    if "attn" in which:
        B, H, D, L = 16, 32, 128, 682
        q = torch.randn(B, L, H, D, device="cuda").to(BF)
        k = torch.randn(B, H, 1024, D, device="cuda").to(BF)
        v = torch.randn(B, H, 1024, D, device="cuda").to(BF)
        out = torch.empty(B * L, H * D, dtype=BF, device="cuda")
        from modelcompose_amd import _lib
        for dbg, nm in ((1, "64-query workgroups"), (2, "128-query workgroups")):
            _lib.lib().mc_attn_debug(dbg)
            t = timeit(lambda: ops.attn_prefill(q, k, v, out, B, H, H, L, L, D, (L*H*D, H*D, D), (H*1024*D, D, 1024*D), (H*1024*D, D, 1024*D), H*D, True))
            fl = 4.0 * L * L * D * H * B / 2
            print(f"attn prefill B={B} L={L} {nm}: {t*1e6:9.1f} us {fl/t/1e12:7.1f} TFLOP/s (causal-counted)")
        _lib.lib().mc_attn_debug(0)
        Ll = 2792
        ql = torch.randn(4, Ll, H, D, device="cuda").to(BF)
        kl = torch.randn(4, H, 2816, D, device="cuda").to(BF)
        vl = torch.randn(4, H, 2816, D, device="cuda").to(BF)
        ol = torch.empty(4 * Ll, H * D, dtype=BF, device="cuda")
        for dbg, nm in ((1, "64-query workgroups"), (2, "128-query workgroups")):
            _lib.lib().mc_attn_debug(dbg)
            t = timeit(lambda: ops.attn_prefill(ql, kl, vl, ol, 4, H, H, Ll, Ll, D, (Ll*H*D, H*D, D), (H*2816*D, D, 2816*D), (H*2816*D, D, 2816*D), H*D, True))
            fl = 4.0 * Ll * Ll * D * H * 4 / 2
            print(f"attn prefill B=4 L={Ll} {nm}: {t*1e6:9.1f} us {fl/t/1e12:7.1f} TFLOP/s (causal-counted)")
        _lib.lib().mc_attn_debug(0)
        for S in (700, 3400):
            Smax = S + 128
            kc = torch.randn(B, H, Smax, D, device="cuda").to(BF)
            vc = torch.randn(B, H, Smax, D, device="cuda").to(BF)
            q1 = torch.randn(B, H, D, device="cuda").to(BF)
            o1 = torch.empty(B, H * D, dtype=BF, device="cuda")
            lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
            for ns in (1, 2, 4, 8):
                ws = torch.empty(B * H * ns * (D + 2), dtype=torch.float32, device="cuda")
                t = timeit(lambda: ops.attn_decode(q1, kc, vc, o1, B, H, H, Smax, D, (H*D, D), (H*Smax*D, D, Smax*D), (H*Smax*D, D, Smax*D), H*D, nsplit=ns, workspace=ws, kv_lens=lens))
                by = 2.0 * B * H * S * D * 2
                print(f"attn decode B={B} S={S} nsplit={ns}: {t*1e6:9.1f} us {by/t/1e12:6.2f} TB/s")


if __name__ == "__main__":
    main()
