"""Decode-attention launches at the bench's decode shape (B 48 sequences x 32 heads x 128, ~2800 cached keys, fused RoPE + append) over 8
rotating KV caches (18 GB: nothing is an Infinity-Cache hit) for a rocprofv3 --pmc FETCH_SIZE pass (tools/pmc_decode_attn.sh).
Prints the algorithmic bytes of one launch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import ops

BF = torch.bfloat16
B, H, D, S, Smax = 48, 32, 128, 2809, 2880
caches = [(torch.randn(B, H, Smax, D, device="cuda").to(BF), torch.randn(B, H, Smax, D, device="cuda").to(BF)) for _ in range(8)]
qkv = torch.randn(B, 3 * H * D, device="cuda").to(BF)
o1 = torch.empty(B, H * D, dtype=BF, device="cuda")
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
ang = torch.outer(torch.arange(4096, dtype=torch.float32), 1.0 / (10000 ** (torch.arange(0, D, 2, dtype=torch.float32) / D)))
cos, sin = ang.cos().cuda().contiguous(), ang.sin().cuda().contiguous()
for it in range(24):
    kc, vc = caches[it % 8]
    ops.attn_decode_rope(qkv, cos, sin, kc, vc, o1, lens, B, H, H, Smax, D)
torch.cuda.synchronize()
print(json.dumps({"decode_attention": {"B": B, "H": H, "D": D, "keys": S, "algorithmic_bytes": 2.0 * B * H * S * D * 2}}))
