"""Where the composition kernel's time goes: one linear (N = K = 4096, r = 128) with 0 .. 6 LoRA terms, 1 .. 4 outputs, with / without the
retention statistic and the column factor; HIP-event time per launch, effective GB/s over the algorithmic bytes (W once + every output)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from modelcompose_amd import ops
from modelcompose_amd.model.multimodal_llama import _compose_multi_into

BF = torch.bfloat16
N = K = 4096
r = 128
g = torch.Generator(device="cuda").manual_seed(1)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(BF)
terms = [((torch.randn(r, K, device="cuda", generator=g) * K ** -0.5).to(BF), (torch.randn(N, r, device="cuda", generator=g) * 0.01).to(BF), 0.5) for _ in range(6)]
cs = (1.0 + 0.1 * torch.randn(K, device="cuda", generator=g)).float()
rows = []
for name, masks, use_ret, use_cs in (("copy_1out_0terms", [0], False, False), ("1out_1term", [1], False, False), ("1out_3terms", [7], False, False),
                                     ("4out_6terms", [7, 8, 16, 32], False, False), ("4out_6terms_colscale", [7, 8, 16, 32], False, True),
                                     ("4out_6terms_colscale_retention", [7, 8, 16, 32], True, True), ("4out_0terms", [0, 0, 0, 0], False, False)):
    outs = [torch.empty(ops.packed_elems(N, K), dtype=BF, device="cuda") for _ in masks]
    rets = [[] for _ in masks] if use_ret else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        _compose_multi_into(w, terms, masks, N, K, outs, col_scale=cs if use_cs else None, retentions=[[] for _ in masks] if use_ret else None)
    torch.cuda.synchronize()
    ev = []
    for _ in range(10):
        _compose_multi_into(w, terms, masks, N, K, outs, col_scale=cs if use_cs else None, retentions=[[] for _ in masks] if use_ret else None, events=ev)
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[len(ev) // 2]
    byts = 2.0 * N * K * (1 + len(masks))
    rows.append({"case": name, "outputs": len(masks), "median_us": round(us, 1), "GBs": round(byts / us / 1e3, 1)})
    print(json.dumps(rows[-1]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/compose_probe.json", "w"), indent=1)
