import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import ops
BF = torch.bfloat16
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
B, H, D = 16, 32, 128
for L in (682, 704, 1364):
    q = torch.randn(B, L, H, D, device="cuda").to(BF); S = (L + 63) // 64 * 64
    k = torch.randn(B, H, S, D, device="cuda").to(BF); v = torch.randn(B, H, S, D, device="cuda").to(BF)
    out = torch.empty(B * L, H * D, dtype=BF, device="cuda")
    for causal in (True, False):
        t = timeit(lambda: ops.attn_prefill(q, k, v, out, B, H, H, L, L, D, (L*H*D, H*D, D), (H*S*D, D, S*D), (H*S*D, D, S*D), H*D, causal))
        fl = 4.0 * L * L * D * H * B / (2 if causal else 1)
        print(f"L={L} causal={causal}: {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TFLOP/s")

print("runtime-like: Lq = kv_lens = 683 inside a 768-row cache (S = 768)")
L, S = 683, 768
q = torch.randn(B, L, H, D, device="cuda").to(BF)
k = torch.randn(B, H, S, D, device="cuda").to(BF); v = torch.randn(B, H, S, D, device="cuda").to(BF)
out = torch.empty(B * L, H * D, dtype=BF, device="cuda")
lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
for tag, kw in (("kv_lens=683", dict(kv_lens=lens)), ("no kv_lens (S=768 all valid)", dict())):
    t = timeit(lambda: ops.attn_prefill(q, k, v, out, B, H, H, L, S, D, (L*H*D, H*D, D), (H*S*D, D, S*D), (H*S*D, D, S*D), H*D, True, **kw))
    fl = 4.0 * L * L * D * H * B / 2
    print(f"  {tag}: {t*1e6:8.1f} us  {fl/t/1e12:6.1f} TFLOP/s")
for L2 in (640, 704, 768):
    q2 = torch.randn(B, L2, H, D, device="cuda").to(BF); out2 = torch.empty(B * L2, H * D, dtype=BF, device="cuda")
    t = timeit(lambda: ops.attn_prefill(q2, k, v, out2, B, H, H, L2, S, D, (L2*H*D, H*D, D), (H*S*D, D, S*D), (H*S*D, D, S*D), H*D, True))
    print(f"  Lq={L2} in the 768-row cache: {t*1e6:8.1f} us  {4.0*L2*L2*D*H*B/2/t/1e12:6.1f} TFLOP/s")

print("Lq sweep in the 768-row cache, kv_lens = Lq")
for L2 in (641, 656, 672, 683, 688, 696, 703, 704):
    q2 = torch.randn(B, L2, H, D, device="cuda").to(BF); out2 = torch.empty(B * L2, H * D, dtype=BF, device="cuda")
    lens2 = torch.full((B,), L2, dtype=torch.int32, device="cuda")
    t = timeit(lambda: ops.attn_prefill(q2, k, v, out2, B, H, H, L2, S, D, (L2*H*D, H*D, D), (H*S*D, D, S*D), (H*S*D, D, S*D), H*D, True, kv_lens=lens2))
    print(f"  Lq={L2}: {t*1e6:8.1f} us  {4.0*L2*L2*D*H*B/2/t/1e12:6.1f} TFLOP/s")
