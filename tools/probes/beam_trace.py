"""Beam search on the device, step by step: the logits every beam row is scored with (cached decode steps over replicated / gathered KV rows)
against a fresh prefill of the same rows through forward(), and the margins of the candidates around rank k (EOS decisions)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from conftest import load_golden
from modelcompose_amd.model.builder import build_from_state_dict

a, meta, sd = load_golden("g4_e2e_vision")
model = build_from_state_dict(meta, sd)
g = torch.Generator().manual_seed(7)
txt = torch.cat([torch.ones(3, 1, dtype=torch.long), torch.randint(3, 97, (3, 7), generator=g)], 1)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
model._beam_trace = []
out = model.generate(txt.cuda(), modal_inputs={}, num_beams=k, max_new_tokens=6, length_penalty=1.0).cpu()
print("out", out.tolist())
V = model.config.vocab_size
for step, (ids, lg) in enumerate(model._beam_trace):
    fresh = model.forward(input_ids=ids.cuda(), modal_inputs={}).logits[:, -1].float().cpu()
    scale = fresh.abs().max()
    d = (lg - fresh).abs().max(dim=1).values / scale
    print(f"step {step}: rows {ids.shape[0]} len {ids.shape[1]}  max |decode - fresh prefill| / scale per row: {[round(float(x), 5) for x in d]}")
    for b in range(ids.shape[0]):
        if float(d[b]) > 5e-3:
            print("   row", b, ids[b].tolist(), "argmax decode", int(lg[b].argmax()), "fresh", int(fresh[b].argmax()))
