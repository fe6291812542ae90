"""GPU-box tool: confirm row seeds for tests/fullwidth_cases.py on the device.

    python tools/screen_rows_gpu.py <case> <seed> [<seed> ...]

Builds the case's HIP model once, runs the given row seeds as one batch, runs both CPU oracles, and prints per row whether all 17 greedy
ids equal the fp32 branch-form oracle's and the device-rounding restatement's, the first differing step and the oracle margin there.
Kernels are deterministic: a row that matches here matches in the test."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fullwidth_cases as fc  # noqa: E402
from oracle import pipeline, splice  # noqa: E402


def main():
    from modelcompose_amd.model.builder import build_from_state_dict
    name = sys.argv[1]
    seeds = [int(a) for a in sys.argv[2:]]
    meta, sd, ids, mi = fc.build_case(name, seeds)
    model = build_from_state_dict(meta, sd)
    if "point" in mi:
        model.model.modal_encoders["point"].fps_start = torch.zeros(ids.shape[0], dtype=torch.long)
    mid = fc.to_dev(mi)
    feats, _ = model.encode_modal_inputs(mid, model.prefix_tokens, model.suffix_tokens)
    res, lg = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True)
    got = res[:, ids.shape[1]:].cpu()
    fb = {m: f.float().cpu() for m, f in feats.items()}
    del model
    torch.cuda.empty_cache()
    sdf = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        od = pipeline.OracleModel.from_state_dict(sdf, meta, emulate="device")
        ids_d, lg_d = od.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True, feats_blocks=fb)
        od._dw = None
        o32 = pipeline.OracleModel.from_state_dict(sdf, meta)
        ids_r, lg_r = o32.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True)
    for b, seed in enumerate(seeds):
        line = [f"row seed {seed}:"]
        ok = True
        for key, io, lo in (("fp32", ids_r, lg_r), ("device", ids_d, lg_d)):
            neq = (got[b] != io[b]).nonzero()
            mm = fc.margins(lo[b:b + 1]).min().item() * (lo[b:b + 1].abs().max() / lo.abs().max()).item()
            if len(neq):
                t = int(neq[0])
                top2 = lo[b, t].topk(2).values
                line.append(f"{key}: first diff at step {t} (oracle margin {(top2[0] - top2[1]).item() / lo.abs().max().item():.4f})")
                ok = False
            else:
                line.append(f"{key}: all equal (min margin {mm:.4f})")
        print(" ".join(line), "  <== MATCH" if ok else "", flush=True)


if __name__ == "__main__":
    main()
