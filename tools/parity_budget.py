"""Error budget of the HIP path's bf16 storage points at the REAL widths (VERDICT r3 #2b/c) - CPU only, build container.

The device-rounding restatement (oracle/device_path.py) with its rounding points switched one at a time, on the 8-layer cut of the
metric's model (tests/fullwidth_cases.py DEPTH_CASES["depth8_iav"]: 3-way composed Vicuna-7B widths, image + audio + video, spliced
length 2793, one row), prefill + 4 teacher-forced decode steps, against the same restatement with NO rounding at all (= the fp32
reference function on pre-merged weights).  For every point: the error when ONLY that point rounds, and the error when every point
but that one rounds (leave-one-out).  `fp32_residual_stream`: both residual sums kept in fp32 with the GEMM operands still rounded to
bf16 - the A/B the verdict asked for.  Distances are max |logit difference| / max |reference logit| and rms / rms.

    python tools/parity_budget.py [layers] > profiles/r04_rounding_budget.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import fullwidth_cases as fc  # noqa: E402
from oracle import device_path, pipeline  # noqa: E402


def main():
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n_new = 5
    torch.set_num_threads(os.cpu_count() or 1)
    name = "depth8_iav"
    fc.DEPTH_CASES[name] = dict(fc.DEPTH_CASES[name], layers=layers)
    meta, sd, ids, mi = fc.build_case(name)
    for k in list(sd):
        if sd[k].is_floating_point() and not k.startswith("model.layers.") and k != "lm_head.weight":
            sd[k] = sd[k].float()
    P = device_path.ROUNDING_POINTS
    off = {k: False for k in P}
    on = {k: True for k in P}

    def run(rounding, blocks=None, forced=None):
        om = pipeline.OracleModel.from_state_dict(sd, meta, emulate="device", device_opts={"lazy": True, "rounding": rounding})
        with torch.no_grad():
            if blocks is None:
                fns = {m: (lambda x, m=m: om.encode_modal(m, x)) for m in om.modals}
                from oracle import splice
                blocks, _ = splice.encode_modal_inputs(fc.to_f32(mi), om.modals, fns, om.prefix, om.suffix, skip_absent=True)
            ids_o, lg = om.generate(ids, fc.to_f32(mi), max_new_tokens=n_new, ignore_eos=True, return_logits=True, feats_blocks=blocks,
                                    forced_ids=forced)
        return ids_o, lg, blocks

    t0 = time.time()
    ids_ref, ref, blocks = run(off)
    scale, rms_ref = ref.abs().max(), ref.pow(2).mean().sqrt()
    dist = lambda lg: {"max": ((lg - ref).abs().max() / scale).item(), "rms": ((lg - ref).pow(2).mean().sqrt() / rms_ref).item()}
    out = {"case": name, "layers": layers, "rows": int(ids.shape[0]), "steps": n_new, "spliced_length": 2793, "logit_scale": scale.item(),
           "reference": "oracle/device_path.py with every rounding point off (fp32 arithmetic on the pre-merged one-adapter-per-token form)",
           "seconds_per_run": round(time.time() - t0, 1), "only": {}, "all_but": {}}
    _, lg_all, _ = run(on, blocks, ids_ref)
    out["all_points"] = dist(lg_all)
    print(f"all points: {out['all_points']}", file=sys.stderr, flush=True)
    for p in P:
        if p == "operand":
            continue                                     # only meaningful with an fp32 stream (below)
        only = dict(off)
        only[p] = True
        _, lg, _ = run(only, blocks, ids_ref)
        out["only"][p] = dist(lg)
        but = dict(on)
        but[p] = False
        _, lg, _ = run(but, blocks, ids_ref)
        out["all_but"][p] = dist(lg)
        print(f"{p}: only {out['only'][p]}  all-but {out['all_but'][p]}", file=sys.stderr, flush=True)
    # fp32 residual stream: both residual sums unrounded, GEMM operands bf16 (what a device implementation would do)
    fp32_stream = dict(on, resid_attn=False, resid_mlp=False, operand=True)
    _, lg, _ = run(fp32_stream, blocks, ids_ref)
    out["fp32_residual_stream"] = dist(lg)
    # and with the operands unrounded as well (the ideal; not implementable: MFMA operands are bf16)
    _, lg, _ = run(dict(on, resid_attn=False, resid_mlp=False, operand=False), blocks, ids_ref)
    out["fp32_residual_stream_and_operands"] = dist(lg)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
