"""Encoder-shaped attention (head_dim 64, bidirectional): LanguageBind-Video spatial (128 frames x 16 heads x 257 tokens), CLIP-L (16 x 16 x 577)."""
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
H, D = 16, 64
for (B, L) in ((128, 257), (16, 577), (4112, 8)):
    q = torch.randn(B, L, H, D, device="cuda").to(BF); k = torch.randn(B, L, H, D, device="cuda").to(BF); v = torch.randn(B, L, H, D, device="cuda").to(BF)
    outs = {}
    for rep in range(2):
        for dbg, nm in (((32, "flash kernel"), (0, "tiny kernel")) if L <= 8 else ((0, "1 block / wave"), (16, "2 blocks / wave"))):
            L_.mc_attn_debug(dbg)
            out = torch.empty(B * L, H * D, dtype=BF, device="cuda")
            st = (L * H * D, H * D, D)
            f = lambda: ops.attn_prefill(q, k, v, out, B, H, H, L, L, D, st, st, st, H * D, False)
            t = timeit(f)
            outs[dbg] = out.clone()
            fl = 4.0 * L * L * D * H * B
            print(f"B={B} L={L} {nm:16s}: {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TFLOP/s")
    ks = sorted(outs)
    ref = torch.softmax(torch.einsum("blhd,bmhd->bhlm", q.float(), k.float()) / 8.0, -1)
    ref = torch.einsum("bhlm,bmhd->blhd", ref, v.float()).reshape(B * L, H * D)
    print("   bit-identical:", torch.equal(outs[ks[0]], outs[ks[1]]), " max err vs fp32:", [round((outs[kk].float() - ref).abs().max().item(), 5) for kk in ks])
L_.mc_attn_debug(0)
