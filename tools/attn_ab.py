"""A/B of the prefill attention shapes (mc_attn_debug bits: 4 = one query block per wave (round-1 kernel), 0 = two blocks per wave)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import _lib, ops
BF = torch.bfloat16
L_ = _lib.lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
H, D = 32, 128
for (B, L) in ((16, 683), (16, 2793), (48, 2793), (4, 3317), (2, 683)):
    S = (L + 32 + 63) // 64 * 64
    q = torch.randn(B, L, H, D, device="cuda").to(BF)
    k = torch.randn(B, H, S, D, device="cuda").to(BF); v = torch.randn(B, H, S, D, device="cuda").to(BF)
    lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
    outs = {}
    for rep in range(2):
        for dbg, nm in ((4, "1 block / wave"), (0, "2 blocks / wave"), (2, "2 blocks, 8 waves"), (8, "2 blocks, 4 waves"), (64, "2 blocks, DMA at the top")):
            L_.mc_attn_debug(dbg)
            out = torch.empty(B * L, H * D, dtype=BF, device="cuda")
            f = lambda: ops.attn_prefill(q, k, v, out, B, H, H, L, S, D, (L*H*D, H*D, D), (H*S*D, D, S*D), (H*S*D, D, S*D), H*D, True, kv_lens=lens)
            t = timeit(f)
            outs[dbg] = out.clone()
            fl = 4.0 * L * L * D * H * B / 2
            print(f"B={B} L={L} {nm:18s}: {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TFLOP/s ({fl/t/2.5e15:.3f} of peak)")
    ref = outs[4].float()
    for dbg in (0, 2, 8, 64):
        d = (outs[dbg].float() - ref).abs().max().item()
        print(f"   max |diff| vs 1-block kernel (dbg {dbg}): {d:.3e}  bit-identical={torch.equal(outs[dbg], outs[4])}")
L_.mc_attn_debug(0)
