"""Fused (RoPE + append) vs plain decode attention at the bench's decode shape, rotating over 8 KV caches (no Infinity-Cache hits)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import ops
BF = torch.bfloat16
B, H, D, S, Smax = 16, 32, 128, 700, 768
caches = [(torch.randn(B, H, Smax, D, device="cuda").to(BF), torch.randn(B, H, Smax, D, device="cuda").to(BF)) for _ in range(8)]
qkv = torch.randn(B, 3 * H * D, device="cuda").to(BF)
q1 = torch.randn(B, H, D, device="cuda").to(BF)
o1 = torch.empty(B, H * D, dtype=BF, device="cuda")
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
half = D // 2
ang = torch.outer(torch.arange(4096, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D)))
cos, sin = ang.cos().cuda().contiguous(), ang.sin().cuda().contiguous()
def timeit(fn, n=400):
    for _ in range(16): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(16): fn()
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n // 16): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (n // 16 * 16)
i = [0]
def plain():
    i[0] = (i[0] + 1) % 8; kc, vc = caches[i[0]]
    ops.attn_decode(q1, kc, vc, o1, B, H, H, Smax, D, (H*D, D), (H*Smax*D, D, Smax*D), (H*Smax*D, D, Smax*D), H*D, nsplit=1, kv_lens=lens)
def fused():
    i[0] = (i[0] + 1) % 8; kc, vc = caches[i[0]]
    ops.attn_decode_rope(qkv, cos, sin, kc, vc, o1, lens, B, H, H, Smax, D)
by = 2.0 * B * H * S * D * 2
for nm, fn in (("plain", plain), ("fused RoPE + append", fused), ("plain", plain), ("fused RoPE + append", fused)):
    t = timeit(fn)
    print(f"decode attention B={B} S={S} {nm:22s}: {t*1e6:7.2f} us  {by/t/1e12:5.2f} TB/s")
