"""The four decode GEMMs of a layer (48 rows against the Vicuna-7B shapes, the epilogues the decode step uses) over 8 rotating weights each
(nothing is an Infinity-Cache hit) for a rocprofv3 --pmc FETCH_SIZE pass (tools/pmc_decode_chain.sh).  Prints the launch order and the
algorithmic bytes (weights once) of every launch of gemm_strip_kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import ops

BF = ops.BF16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 48
order = []
for (name, N, K, kw) in (("qkv", 12288, 4096, dict(rms_eps=1e-5)), ("o", 4096, 4096, dict(res=True)), ("gate_up", 22016, 4096, dict(rms_eps=1e-5, swiglu=True)),
                         ("down", 4096, 11008, dict(res=True))):
    ws = [ops.pack_weight(torch.randn(N, K, device="cuda", dtype=BF) * 0.02) for _ in range(8)]
    x = torch.randn(M, K, device="cuda", dtype=BF)
    out = torch.randn(M, N // 2 if kw.get("swiglu") else N, device="cuda", dtype=BF)
    for it in range(24):
        ops.linear_ex(x, ws[it % 8], residual=out if kw.get("res") else None, out=out, swiglu=bool(kw.get("swiglu")), rms_eps=kw.get("rms_eps", 0.0))
        order.append({"name": name, "algorithmic_bytes": 2.0 * N * K, "x_bytes": 2.0 * M * K})
    torch.cuda.synchronize()
    del ws
print(json.dumps({"decode_chain": {"rows": M, "launch_order": order}}))
