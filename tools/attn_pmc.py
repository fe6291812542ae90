"""One causal prefill-attention launch set at the headline shape (B 16, L 2793, 32 heads x 128) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import ops
BF = torch.bfloat16
B, L, H, D = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 2793, 32, 128
S = (L + 32 + 63) // 64 * 64
q = torch.randn(B, L, H, D, device="cuda").to(BF)
k = torch.randn(B, H, S, D, device="cuda").to(BF); v = torch.randn(B, H, S, D, device="cuda").to(BF)
lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
out = torch.empty(B * L, H * D, dtype=BF, device="cuda")
for _ in range(4):
    ops.attn_prefill(q, k, v, out, B, H, H, L, S, D, (L*H*D, H*D, D), (H*S*D, D, S*D), (H*S*D, D, S*D), H*D, True, kv_lens=lens)
torch.cuda.synchronize()
