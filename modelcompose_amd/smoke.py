"""One tiny composed-model generation on cuda:0, checked against the CPU oracle (called by __graft_entry__.smoke())."""
from __future__ import annotations

import json
import os

import numpy as np
import torch


def _load_fixture(name):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", name + ".npz"), allow_pickle=False)
    arrays, sd, meta = {}, {}, None
    for k in z.files:
        if k == "meta":
            meta = json.loads(str(z[k]))
        elif k.startswith("sd::"):
            sd[k[4:]] = torch.from_numpy(z[k])
        else:
            arrays[k] = torch.from_numpy(z[k])
    return arrays, meta, sd


def run():
    from modelcompose_amd.model.builder import build_from_state_dict
    from oracle import pipeline          # checker only
    a, meta, sd = _load_fixture("g4_e2e_vision")
    model = build_from_state_dict(meta, sd)
    ids, px = a["input_ids"].cuda(), a["pixels"].cuda()
    res, lg = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=4, ignore_eos=True, return_step_logits=True)
    torch.cuda.synchronize()
    om = pipeline.OracleModel.from_state_dict(sd, meta)
    ids_o, lg_o = om.generate(a["input_ids"], {"vision": a["pixels"]}, max_new_tokens=4, ignore_eos=True, return_logits=True)
    err = (lg.float().cpu() - lg_o).abs().max().item() / lg_o.abs().max().item()
    assert err < 1e-2, f"smoke: logits differ from the oracle by {err:.3e} of the logit scale (measured 5.0e-3 on MI355X)"
    assert torch.equal(res[:, ids.shape[1]:].cpu(), ids_o), "smoke: greedy ids differ from the oracle"
    print(f"smoke ok: rel logit err {err:.2e}; ids {res[:, ids.shape[1]:].tolist()} oracle {ids_o.tolist()}")
