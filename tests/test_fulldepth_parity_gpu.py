"""Parity at the DEPTH that is benchmarked: the metric's model - 3-way composed Vicuna-7B (online-merge-reset
vision / audio / video = 0.333, routed adapters default / audio / vision / video), all 32 decoder layers at the real widths, real-size
CLIP-L/336 + BEATs/Q-Former + LanguageBind-Video encoders - on two UNSCREENED rows of the metric's inputs (336 px image + 10 s audio +
8-frame video, spliced length 2793), prefill + 16 greedy decode steps through the C ABI, against the committed output of the pinned
fp32 branch-form oracle (tests/golden/g15_fulldepth_iav.npz, written in the build container by `python -m oracle.gen_golden g15`:
35 GB of fp32 weights, ~15 minutes on 8 cores - too much to run beside the test).  A second fixture holds the same model cut to
8 layers, so the error's growth with depth (2 -> 8 -> 32 layers) is measured, not extrapolated.

Reference: modelcompose/model/language_model/multimodal_llama.py:488-619 (model forward), :676-767 (lm_head / generation inputs),
eval/model_multimodal_qa_loader.py:94-108 (the greedy generate call the metric times).

What is asserted (tests/test_fullwidth_parity_gpu.py explains why a fixed 1e-3 cannot hold end to end with bf16 storage; measured
on MI355X the distance from the fp32 oracle grows 1.1e-2 -> 1.8e-2 -> 3.0e-2 of the logit scale over 2 -> 8 -> 32 layers, i.e. like the
square root of the depth, as independent per-layer rounding noise does):
  * the inputs generated on this box are the ones the fixture was made from;
  * TEACHER-FORCED pass (generate(forced_ids = the oracle's ids)): the logits of ALL 17 steps - prefill and 16 cached decode steps, each
    conditioned on the oracle's own history - within DEPTH_BOUND = 2x the error measured on MI355X (of the oracle's logit scale), and
    the device's argmax at every step equal to the oracle's token unless the oracle's top-2 margin at that step is inside the band;
  * FREE pass (the shipped loop, hipGraph replay): greedy ids equal to the oracle's up to the first step whose oracle top-2 margin is
    inside the band; at that step the device's token must be one the oracle itself ranks within the band of its best.  (With a logit
    error of 3e-2 against a mean top-2 gap of 4.4e-2 for this random-weight model, a 32-layer row leaves the oracle's path within
    the first few tokens more often than not - both rows of the fixture do so at token 0, at oracle margins 7.5e-3 and 1.7e-3 - which
    is why the teacher-forced pass carries the per-step comparison.)
"""
import json
import os

import numpy as np
import pytest
import torch

import fullwidth_cases as fc

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# 2x the max |logit error| / max |oracle logit| measured on MI355X in round 3 (profiles/r03_parity.json)
# 2x the max |logit error| / max |oracle logit| over the 17 teacher-forced steps measured on MI355X in round 3 (1.76e-2 / 4.01e-2)
DEPTH_BOUND = {"depth8_iav": 3.6e-2, "fulldepth_iav": 8.0e-2}
# a device argmax may differ from the oracle's only where the oracle's top-2 margin is below the measured size of the error itself
# (observed departures: margins 1.7e-3 ... 1.04e-2)
NEAR_TIE = {"depth8_iav": 1.8e-2, "fulldepth_iav": 4.0e-2}
REPORT = {}


def _fixture(fname):
    path = os.path.join(GOLD, fname + ".npz")
    if not os.path.exists(path):
        pytest.fail(f"{path} is missing: python -m oracle.gen_golden g15 (build container)")
    z = np.load(path)
    return {k: z[k] for k in z.files}


def _run(name, fname):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd.model.builder import build_from_state_dict
    gold = _fixture(fname)
    info = json.loads(bytes(gold["meta"]).decode())
    assert info["case"] == name and info["row_seeds"] == fc.DEPTH_CASES[name]["row_seeds"] and info["seed"] == fc.DEPTH_CASES[name]["seed"]
    meta, sd, ids, mi = fc.build_case(name)
    assert meta["num_hidden_layers"] == info["layers"]
    assert np.array_equal(ids.numpy(), gold["input_ids"]), "this box generated other prompts than the fixture's"
    model = build_from_state_dict(meta, sd)
    del sd
    mid = fc.to_dev(mi)
    ref_ids, ref_lg = torch.from_numpy(gold["ids"]), torch.from_numpy(gold["logits"])
    res, lg = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True)
    res_graph = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True)          # shipped path: hipGraph replay
    graph_active = model.runtime_option("graph_active")
    got_ids, got_lg = res[:, ids.shape[1]:].cpu(), lg.float().cpu()
    assert torch.equal(res_graph[:, ids.shape[1]:].cpu(), got_ids) and graph_active == 1
    # teacher forcing: decode step s is fed the oracle's token s
    res_tf, lg_tf = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True,
                                   forced_ids=ref_ids[:, :fc.N_NEW - 1])
    tf_ids, tf_lg = res_tf[:, ids.shape[1]:].cpu(), lg_tf.float().cpu()
    assert torch.equal(tf_lg[:, 0], got_lg[:, 0])                    # the prefill step is the same launch sequence in both passes
    del model
    torch.cuda.empty_cache()
    assert got_lg.shape == ref_lg.shape == (ids.shape[0], fc.N_NEW, meta["vocab_size"])
    scale = ref_lg.abs().max()
    bound = DEPTH_BOUND[name]
    rows = []
    worst = 0.0
    for b in range(ids.shape[0]):
        neq = (got_ids[b] != ref_ids[b]).nonzero()
        t = int(neq[0]) if len(neq) else fc.N_NEW
        upto = min(t + 1, fc.N_NEW)                                 # steps that saw the oracle's token history
        err = ((got_lg[b, :upto] - ref_lg[b, :upto]).abs().max() / scale).item()
        err0 = ((got_lg[b, 0] - ref_lg[b, 0]).abs().max() / scale).item()
        rms = ((got_lg[b, :upto] - ref_lg[b, :upto]).pow(2).mean().sqrt() / ref_lg[b, :upto].pow(2).mean().sqrt()).item()
        worst = max(worst, err)
        row = {"row": b, "steps_on_the_oracle_path": t, "max_err_over_those_steps": err, "prefill_step_err": err0, "rms_err_over_rms_logit": rms,
               "min_oracle_margin_over_those_steps": fc.margins(ref_lg[b:b + 1, :upto]).min().item() if upto else None}
        if t < fc.N_NEW:
            top2 = ref_lg[b, t].topk(2)
            row["departure_margin"] = ((top2.values[0] - top2.values[1]) / scale).item()
            row["departure_chosen_gap"] = ((top2.values[0] - ref_lg[b, t, int(got_ids[b, t])]) / scale).item()
        rows.append(row)
    # teacher-forced: every step is comparable
    tf_err = ((tf_lg - ref_lg).abs().amax(-1) / scale)                                    # (B, 17)
    tf_rms = ((tf_lg - ref_lg).pow(2).mean(-1).sqrt() / ref_lg.pow(2).mean(-1).sqrt())
    marg = fc.margins(ref_lg)                                                             # oracle top-2 gaps / logit scale, (B, 17)
    agree = tf_ids == ref_ids
    tf = {"max_err": tf_err.max().item(), "max_err_per_step": tf_err.amax(0).tolist(), "rms_err_over_rms_logit": tf_rms.mean().item(),
          "argmax_agrees": int(agree.sum()), "argmax_total": int(agree.numel()),
          "disagreements": [{"row": int(b), "step": int(t), "oracle_margin": marg[b, t].item(),
                             "chosen_gap": ((ref_lg[b, t].max() - ref_lg[b, t, int(tf_ids[b, t])]) / scale).item()}
                            for b, t in (~agree).nonzero().tolist()]}
    worst = max(worst, tf["max_err"])
    REPORT[name] = {"layers": info["layers"], "teacher_forced": tf, "free_rows": rows, "max_err": worst, "bound": bound, "logit_scale": scale.item(),
                    "spliced_length": 2793, "oracle": "fp32 branch form (oracle/pipeline.py), fixture " + fname}
    out = os.path.join(fc.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(REPORT, open(os.path.join(out, "fulldepth_parity.json"), "w"), indent=1)
    print(name, json.dumps(REPORT[name]))
    assert tf["max_err"] <= bound, tf
    tie = NEAR_TIE[name]
    for d in tf["disagreements"]:
        assert d["oracle_margin"] < tie and d["chosen_gap"] < tie, d
    for row in rows:
        assert row["max_err_over_those_steps"] <= bound, row
        if row["steps_on_the_oracle_path"] < fc.N_NEW:
            # a departure is legitimate only at an oracle near-tie: margin and the chosen token's gap inside the error band of a logit
            # DIFFERENCE (two logits, each within `bound`/2 typical error)
            assert row["departure_margin"] < tie and row["departure_chosen_gap"] < tie, row
    return rows


def test_eight_layers_against_the_committed_oracle_fixture():
    _run("depth8_iav", "g15_depth8_iav")


def test_full_depth_32_layers_against_the_committed_oracle_fixture():
    rows = _run("fulldepth_iav", "g15_fulldepth_iav")
    # the first generated token (the prefill's argmax over 32 layers x 2793 positions) is the oracle's unless its margin is a near-tie
    for row in rows:
        assert row["steps_on_the_oracle_path"] >= 1 or row["departure_margin"] < NEAR_TIE["fulldepth_iav"]


# ------------------------------------------------------------------------------------------------------------------- round 4: eight rows, both oracles
# tests/golden/g17_fulldepth_iav8.npz (`python -m oracle.gen_golden g17a / g17b / g17`): the fp32 branch-form oracle's free-running ids +
# logits for EIGHT unscreened rows, and the logits of the device-rounding restatement (oracle/device_path.py) teacher-forced on those ids.
# Bounds = 2x what MI355X measured in round 4 (profiles/r04_parity.json).
# HIP backbone vs the device-rounding oracle, both fed the fp32 oracle's encoder blocks, 32 layers: measured 2.87e-2 (rms 3.0e-2) - NOT a
# fraction of the fp32 distance (3.4e-2): the restatement itself sits 3.15e-2 from fp32, and two implementations with the same storage
# points but different fp32 summation order decorrelate to the full bf16 noise level within a layer, so at depth the three
# pairwise distances are all of one size.  What the pair of oracles does pin: the HIP path is no noisier than the CPU restatement (below).
DEV_BOUND = 5.8e-2
FP32_BOUND8 = 8.0e-2        # HIP (own bf16 encoders) vs the fp32 oracle, as for the two-row fixture


def _oracle_feature_blocks(meta, sd, mi):
    """[prefix | projected features | suffix] blocks of the fp32 ORACLE encoders (oracle/pipeline.py), computed on this box's host cores -
    what the fixture's device-rounding logits were made from (the build container computed the same function on other cores: the blocks
    agree to fp32 summation order, far below one bf16 step of the values the backbone receives)."""
    from oracle import pipeline, splice
    keep = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items() if not k.startswith("model.layers.") and k != "lm_head.weight"}
    om = pipeline.OracleModel.from_state_dict(keep, dict(meta, num_hidden_layers=0))
    fns = {m: (lambda x, m=m: om.encode_modal(m, x)) for m in om.modals}
    with torch.no_grad():
        feats, _ = splice.encode_modal_inputs(fc.to_f32(mi), om.modals, fns, om.prefix, om.suffix, skip_absent=True)
    return feats


def test_full_depth_eight_rows_against_both_oracles():
    """32 layers, eight unscreened rows of image + audio + video.
      (1) HIP as shipped (its own bf16 encoders), teacher-forced on the fp32 oracle's ids, vs the fp32 branch-form oracle: logits within
          FP32_BOUND8, argmax disagreements only at oracle near-ties; free-running: tokens matched per row are REPORTED;
      (2) the check that can see a depth-dependent defect of the BACKBONE: HIP fed the fp32 oracle's encoder blocks vs the device-rounding
          restatement (same storage points, same pre-merged weights, CPU fp32 arithmetic) fed the same blocks, same teacher-forced history:
          within DEV_BOUND, and no further from the fp32 oracle than 1.5x the restatement's own distance from it (+5e-3) - a backbone that
          accumulated a depth-dependent error would be noisier than the CPU implementation with the same storage points."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import time
    from modelcompose_amd.model.builder import build_from_state_dict
    name = "fulldepth_iav8"
    gold = _fixture("g17_fulldepth_iav8")
    info = json.loads(bytes(gold["meta"]).decode())
    assert info["case"] == name and info["row_seeds"] == fc.DEPTH_CASES[name]["row_seeds"] and info["layers"] == 32
    meta, sd, ids, mi = fc.build_case(name)
    assert np.array_equal(ids.numpy(), gold["input_ids"]), "this box generated other prompts than the fixture's"
    t0 = time.time()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    blocks = _oracle_feature_blocks(meta, sd, mi)
    t_enc = time.time() - t0
    model = build_from_state_dict(meta, sd)
    del sd
    mid = fc.to_dev(mi)
    ref_ids, ref_lg, dev_lg = torch.from_numpy(gold["ids"]), torch.from_numpy(gold["logits"]), torch.from_numpy(gold["logits_device"])
    B = ids.shape[0]
    forced = ref_ids[:, :fc.N_NEW - 1]
    # (1) as shipped
    res, lg = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True, forced_ids=forced)
    tf_ids, tf_lg = res[:, ids.shape[1]:].cpu(), lg.float().cpu()
    free = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True)[:, ids.shape[1]:].cpu()
    # (2) backbone only: the model's encoders replaced by the oracle's blocks
    own = model.encode_modal_inputs
    dev_blocks = {m: f.to(torch.bfloat16).cuda().contiguous() for m, f in blocks.items()}
    model.encode_modal_inputs = lambda inputs, prefix_tokens=None, suffix_tokens=None: (
        {m: dev_blocks[m] for m in dev_blocks if m in inputs}, {m: torch.ones(dev_blocks[m].shape[:2], device="cuda") for m in dev_blocks if m in inputs})
    try:
        res2, lg2 = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True, forced_ids=forced)
    finally:
        model.encode_modal_inputs = own
    bb_ids, bb_lg = res2[:, ids.shape[1]:].cpu(), lg2.float().cpu()
    del model
    torch.cuda.empty_cache()
    scale = ref_lg.abs().max()
    e_fp32 = (tf_lg - ref_lg).abs().amax(-1) / scale                      # (B, 17) HIP as shipped vs fp32 oracle
    e_dev = (bb_lg - dev_lg).abs().amax(-1) / scale                       # HIP backbone vs device-rounding oracle (same blocks)
    e_bb32 = (bb_lg - ref_lg).abs().amax(-1) / scale                      # HIP backbone (oracle blocks) vs fp32 oracle
    e_o = (dev_lg - ref_lg).abs().amax(-1) / scale                        # the restatement's own distance from fp32
    marg = fc.margins(ref_lg)
    agree = tf_ids == ref_ids
    matched = [int(((free[b] != ref_ids[b]).nonzero()[0]) if (free[b] != ref_ids[b]).any() else fc.N_NEW) for b in range(B)]
    dev_ids = torch.from_numpy(gold["ids_device"])
    rep = {"layers": 32, "rows": B, "row_seeds": info["row_seeds"], "logit_scale": scale.item(), "oracle_encoder_seconds_on_this_host": round(t_enc, 1),
           "hip_vs_fp32_oracle": {"max": e_fp32.max().item(), "per_row_max": e_fp32.amax(1).tolist(), "per_step_max": e_fp32.amax(0).tolist(),
                                  "rms_over_rms_logit": ((tf_lg - ref_lg).pow(2).mean().sqrt() / ref_lg.pow(2).mean().sqrt()).item()},
           "hip_backbone_vs_device_rounding_oracle": {"max": e_dev.max().item(), "per_row_max": e_dev.amax(1).tolist(), "per_step_max": e_dev.amax(0).tolist(),
                                                      "rms_over_rms_logit": ((bb_lg - dev_lg).pow(2).mean().sqrt() / dev_lg.pow(2).mean().sqrt()).item(),
                                                      "argmax_agrees": int((bb_ids == dev_ids).sum()), "argmax_total": int(dev_ids.numel())},
           "hip_backbone_vs_fp32_oracle": {"max": e_bb32.max().item()},
           "device_rounding_oracle_vs_fp32_oracle": {"max": e_o.max().item(), "argmax_agrees": int((dev_ids == ref_ids).sum())},
           "teacher_forced_argmax": {"agrees": int(agree.sum()), "total": int(agree.numel()),
                                     "disagreements": [{"row": int(b), "step": int(t), "oracle_margin": marg[b, t].item(),
                                                        "chosen_gap": ((ref_lg[b, t].max() - ref_lg[b, t, int(tf_ids[b, t])]) / scale).item()}
                                                       for b, t in (~agree).nonzero().tolist()]},
           "free_running_tokens_matched_per_row": matched,
           "free_running_departure_margins": [marg[b, matched[b]].item() if matched[b] < fc.N_NEW else None for b in range(B)],
           "mean_oracle_top2_margin": marg.mean().item()}
    REPORT[name] = rep
    out = os.path.join(fc.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(REPORT, open(os.path.join(out, "fulldepth_parity.json"), "w"), indent=1)
    print(name, json.dumps(rep))
    assert rep["hip_vs_fp32_oracle"]["max"] <= FP32_BOUND8, rep["hip_vs_fp32_oracle"]
    assert rep["hip_backbone_vs_device_rounding_oracle"]["max"] <= DEV_BOUND, rep["hip_backbone_vs_device_rounding_oracle"]
    # no noisier than a CPU implementation with the same storage points: the HIP backbone is no further from fp32 than 1.5x the restatement is
    assert rep["hip_backbone_vs_fp32_oracle"]["max"] <= 1.5 * rep["device_rounding_oracle_vs_fp32_oracle"]["max"] + 5e-3, rep
    tie = NEAR_TIE["fulldepth_iav"]
    for d in rep["teacher_forced_argmax"]["disagreements"]:
        assert d["oracle_margin"] < tie and d["chosen_gap"] < tie, d
    for b in range(B):
        if matched[b] < fc.N_NEW:
            assert rep["free_running_departure_margins"][b] < tie, (b, rep["free_running_departure_margins"][b])
