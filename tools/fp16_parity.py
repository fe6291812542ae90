#!/usr/bin/env python
"""Parity of the HIP path against the committed oracle / reference fixtures in a chosen STORAGE dtype.

    python tools/fp16_parity.py --dtype fp16 --case g4 g8 fulldepth8 [--out gpurun_out/r05_parity_fp16.json]

bf16 (libmc_hip.so) is BASELINE.json's dtype and the headline; fp16 (libmc_hip_f16.so: the same sources instantiated on IEEE half) is the
reference's own inference dtype (modelcompose/model/builder.py:41, :162, :185) with an 8x finer mantissa - the instrument that shows how much of
the distance from the fp32 oracle is bf16 rounding and how much would be a defect.  One storage dtype per process, hence a tool that the
tests start as a child process.

cases:  g4 / g8      the reference's own tiny fixtures (tests/golden): prefill + step logits, greedy ids
        fulldepth8   the metric's model, 32 layers, eight unscreened rows of image + audio + video (tests/golden/g17): logits of 17
                     teacher-forced steps against the fp32 branch-form oracle, argmax agreement, free-running tokens matched per row
Prints one JSON object."""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"])
    ap.add_argument("--case", nargs="+", default=["g4"], choices=["g4", "g8", "fulldepth8"])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import numpy as np
    import torch
    import modelcompose_amd
    from modelcompose_amd import _lib
    modelcompose_amd.set_storage_dtype(a.dtype)
    from modelcompose_amd.model.builder import build_from_state_dict
    from conftest import load_golden
    dt = _lib.storage_dtype()
    rep = {"dtype": a.dtype, "library": os.path.basename(_lib.LIB_PATHS[_lib.storage_name()]), "cases": {}}

    def rel(got, ref):
        return ((got.float().cpu() - ref.float()).abs().max() / ref.float().abs().max()).item()

    def cast(x):
        if isinstance(x, dict):
            return {k: cast(v) for k, v in x.items()}
        return x.to("cuda", dt) if torch.is_tensor(x) and x.is_floating_point() else (x.cuda() if torch.is_tensor(x) else x)

    for case in a.case:
        if case in ("g4", "g8"):
            name = "g4_e2e_vision" if case == "g4" else "g8_e2e_4modal"
            arr, meta, sd = load_golden(name)
            model = build_from_state_dict(meta, sd)
            if case == "g4":
                mi = {"vision": cast(arr["pixels"])}
            else:
                mi = {"vision": cast(arr["pixels"]), "audio": {"audio_inputs": cast(arr["fbank"]), "audio_padding_mask": arr["padding_mask"].cuda()},
                      "video": cast(arr["video"]), "point": cast(arr["points"])}
            ids = arr["input_ids"].cuda()
            out = model.forward(input_ids=ids, modal_inputs=mi)
            n_new = arr["gen_ids"].shape[1]
            res, lg = model.generate(ids, modal_inputs=mi, max_new_tokens=n_new, ignore_eos=True, return_step_logits=True)
            got = res[:, ids.shape[1]:].cpu()
            rep["cases"][case] = {"prefill_logits_err": rel(out.logits, arr["logits_prefill"]), "step_logits_err": rel(lg, arr["step_logits"]),
                                  "ids_equal": bool(torch.equal(got, arr["gen_ids"])), "ids_total": int(got.numel()),
                                  "ids_matching": int((got == arr["gen_ids"]).sum())}
            del model
            torch.cuda.empty_cache()
        else:
            import fullwidth_cases as fc
            z = np.load(os.path.join(ROOT, "tests", "golden", "g17_fulldepth_iav8.npz"))
            name = "fulldepth_iav8"
            meta, sd, ids, mi = fc.build_case(name)
            assert np.array_equal(ids.numpy(), z["input_ids"])
            model = build_from_state_dict(meta, sd)
            del sd
            mid = cast(mi)
            ref_ids, ref_lg = torch.from_numpy(z["ids"]), torch.from_numpy(z["logits"])
            forced = ref_ids[:, :fc.N_NEW - 1]
            res, lg = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True, forced_ids=forced)
            tf_ids, tf_lg = res[:, ids.shape[1]:].cpu(), lg.float().cpu()
            free = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True)[:, ids.shape[1]:].cpu()
            scale = ref_lg.abs().max()
            e = (tf_lg - ref_lg).abs().amax(-1) / scale
            marg = fc.margins(ref_lg)
            agree = tf_ids == ref_ids
            B = ids.shape[0]
            matched = [int((free[b] != ref_ids[b]).nonzero()[0]) if (free[b] != ref_ids[b]).any() else fc.N_NEW for b in range(B)]
            rep["cases"][case] = {
                "layers": 32, "rows": B, "steps": fc.N_NEW, "logit_scale": scale.item(),
                "hip_vs_fp32_oracle": {"max": e.max().item(), "per_row_max": e.amax(1).tolist(), "per_step_max": e.amax(0).tolist(),
                                       "rms_over_rms_logit": ((tf_lg - ref_lg).pow(2).mean().sqrt() / ref_lg.pow(2).mean().sqrt()).item()},
                "teacher_forced_argmax": {"agrees": int(agree.sum()), "total": int(agree.numel()),
                                          "disagreement_margins": [marg[b, t].item() for b, t in (~agree).nonzero().tolist()]},
                "free_running_tokens_matched_per_row": matched,
                "free_running_departure_margins": [marg[b, matched[b]].item() if matched[b] < fc.N_NEW else None for b in range(B)],
                "mean_oracle_top2_margin": marg.mean().item()}
            del model
            torch.cuda.empty_cache()
    line = json.dumps(rep)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(json.dumps(rep, indent=1))
    print(line)


if __name__ == "__main__":
    main()
