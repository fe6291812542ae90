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
        small_delta  the two-layer real-width vision model with LoRA B scaled to |dW| / |W| ~ 2^-9.3 (tests/test_fullwidth_parity_gpu.py::
                     test_small_delta_*): HIP(delta) - HIP(B = 0) projected on the fp32 branch-form oracle's difference, 17 teacher-forced
                     steps - the share of a small delta that arrives, and the rounding noise beside it, in this storage dtype
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
    ap.add_argument("--case", nargs="+", default=["g4"], choices=["g4", "g8", "fulldepth8", "small_delta"])
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
        elif case == "small_delta":
            import warnings
            import fullwidth_cases as fc
            from oracle import pipeline, splice
            meta, sd, ids, mi = fc.build_case("configs1_vision", lora_b_std=0.01 / 64)
            w = sd["model.layers.0.self_attn.q_proj.weight"].float()
            dw = 2.0 * sd["model.layers.0.self_attn.q_proj.lora_B.default.weight"].float() @ sd["model.layers.0.self_attn.q_proj.lora_A.default.weight"].float()
            mid = cast(mi)
            sdf = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
            sd_zero = {k: (torch.zeros_like(v) if ".lora_B." in k else v) for k, v in sd.items()}

            def hip(state, forced=None):
                with warnings.catch_warnings(record=True) as wl:
                    warnings.simplefilter("always")
                    model = build_from_state_dict(meta, state)
                feats, _ = model.encode_modal_inputs(mid, model.prefix_tokens, model.suffix_tokens)
                res, lg = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True,
                                         **({"forced_ids": forced} if forced is not None else {}))
                o = (lg.float().cpu(), {m: f.float().cpu() for m, f in feats.items()},
                     {"final": dict(model.delta_retention), "rne": dict(getattr(model, "delta_retention_rne", {})), "dithered": list(getattr(model, "delta_dithered", []))},
                     [str(w_.message) for w_ in wl if issubclass(w_.category, RuntimeWarning)])
                del model
                torch.cuda.empty_cache()
                return o
            _, fb, _, _ = hip(sd)
            with torch.no_grad():
                o32 = pipeline.OracleModel.from_state_dict(sdf, meta)
                ids_r, lg_r = o32.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True, feats_blocks=fb)
                o0 = pipeline.OracleModel.from_state_dict({k: (torch.zeros_like(v) if ".lora_B." in k else v) for k, v in sdf.items()}, meta)
                _, lg_0 = o0.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True, feats_blocks=fb, forced_ids=ids_r)
            forced = ids_r[:, :fc.N_NEW - 1]
            lg, _, retention, warned = hip(sd, forced)
            lg_h0, _, _, _ = hip(sd_zero, forced)
            dH, dO = (lg - lg_h0).double(), (lg_r - lg_0).double()
            c = ((dH * dO).sum() / (dO * dO).sum()).item()
            rep["cases"][case] = {"dw_over_w": (dw.abs().mean() / w.abs().mean()).item(), "rows": int(ids.shape[0]), "steps": fc.N_NEW,
                                  "delta_projection_hip_on_oracle": c,
                                  "delta_projection_per_step": ((dH * dO).sum((0, 2)) / (dO * dO).sum((0, 2))).tolist(),
                                  "orthogonal_residual_over_effect": ((dH - c * dO).norm() / dO.norm()).item(),
                                  "delta_effect_on_logits": rel(lg_0, lg_r), "logit_err_vs_branch_form": rel(lg, lg_r),
                                  "compose_retention_per_adapter": retention, "warned": warned}
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
