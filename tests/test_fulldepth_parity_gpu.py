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


PROBE_LAYERS = (0, 8, 16, 24, 31)          # per-layer teacher-forced check (below)


@pytest.fixture(scope="module")
def model32():
    """The metric's model at full depth - ONE build for every 32-layer test of this file (fulldepth_iav and fulldepth_iav8 are the same
    weights: seed 41, 3-way composed, 32 layers; they differ in their rows).  Keeps, on the host, what the oracle sides need: the
    non-layer tensors (encoders, embeddings, head) and the tensors of the probed layers."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd.model.builder import build_from_state_dict
    meta, sd = fc.build_weights("fulldepth_iav8")
    keep = {k: v for k, v in sd.items() if not k.startswith("model.layers.") or int(k.split(".")[2]) in PROBE_LAYERS}
    model = build_from_state_dict(meta, sd)
    del sd
    yield model, meta, keep
    del model
    torch.cuda.empty_cache()


def _run(name, fname, prebuilt=None):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd.model.builder import build_from_state_dict
    gold = _fixture(fname)
    info = json.loads(bytes(gold["meta"]).decode())
    assert info["case"] == name and info["row_seeds"] == fc.DEPTH_CASES[name]["row_seeds"] and info["seed"] == fc.DEPTH_CASES[name]["seed"]
    if prebuilt is None:
        meta, sd, ids, mi = fc.build_case(name)
        model = build_from_state_dict(meta, sd)
        del sd
    else:
        model, meta, _ = prebuilt
        ids, mi = fc.build_rows(name)
    assert meta["num_hidden_layers"] == info["layers"]
    assert np.array_equal(ids.numpy(), gold["input_ids"]), "this box generated other prompts than the fixture's"
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
    if prebuilt is None:
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


def test_full_depth_32_layers_against_the_committed_oracle_fixture(model32):
    rows = _run("fulldepth_iav", "g15_fulldepth_iav", prebuilt=model32)
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


def test_full_depth_eight_rows_against_both_oracles(model32):
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
    model, meta, sd = model32
    ids, mi = fc.build_rows(name)
    assert np.array_equal(ids.numpy(), gold["input_ids"]), "this box generated other prompts than the fixture's"
    t0 = time.time()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    blocks = _oracle_feature_blocks(meta, sd, mi)
    t_enc = time.time() - t0
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


# ------------------------------------------------------------------------------------------------------------------- round 5: the benchmarked shape
def _dump():
    out = os.path.join(fc.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(REPORT, open(os.path.join(out, "fulldepth_parity.json"), "w"), indent=1)


def test_fixture_rows_inside_the_benchmarked_48_row_batch(model32):
    """the shape bench.py times - B = 48 rows x 2793 tokens (M = 134 064 routed rows: the slab raster, the XCD round-robin
    raster and the grouped launches of the 256 x 256 kernel are all live), 32 layers, two generation pipelines - had no test.  The eight
    fixture rows (g17) are placed inside two 48-row batches among 40 other rows, at different positions in each.
      (1) EXACT: prefill is batch-invariant - a row's result does not depend on the rows around it or on where its tiles fall.  The
          final-norm hidden state of every token of the fixture rows and their K / V cache entries of all 32 layers are compared BITWISE
          between the 8-row run and each 48-row run (forward()'s route: every layer on every row).  [The logits themselves go through
          lm_head at M = B rows, where the kernel - and so the fp32 summation order - is chosen by B: they are compared in (2).]
      (2) SHIPPED: generate_pipelined (last-layer tail, hipGraph decode, two slots) over the two batches.  Round 6 (VERDICT r5 #1a): decode
          is batch-invariant too - the strip GEMM family adds a row's products in one order whatever M is (csrc/gemm_strip.hip), the
          decode attention's chunk schedule is a function of each sequence's own length (csrc/attention.hip) - so every step's logits of
          the fixture rows are BITWISE the 8-row run's and all 17 greedy tokens are equal, wherever the rows sit in the 48-row batch.
          This is what lets the eval loader batch a question differently at 1 and at 8 GPUs (model_multimodal_qa_loader.py:25-46) and
          still `cat` identical answer files (MCUB-4.sh:60-70)."""
    model, meta, _ = model32
    name = "fulldepth_iav8"
    ids8, mi8 = fc.build_rows(name)
    fill_seeds = list(range(900, 940))
    idsf, mif = fc.build_rows(name, row_seeds=fill_seeds)
    place = [[0, 5, 13, 22, 31, 40, 46, 47], [47, 3, 17, 8, 29, 36, 41, 20]]      # slot of fixture row r in batch 0 / 1

    def cat_rows(a, b, order):
        if isinstance(a, dict):
            return {k: cat_rows(a[k], b[k], order) for k in a}
        return torch.cat([a, b], 0)[order]

    batches = []
    for pl in place:
        others = [i for i in range(48) if i not in pl]
        order = torch.empty(48, dtype=torch.long)
        order[torch.tensor(pl)] = torch.arange(8)
        order[torch.tensor(others)] = 8 + torch.arange(40)
        batches.append((cat_rows(ids8, idsf, order).cuda(), fc.to_dev(cat_rows(mi8, mif, order))))
        assert torch.equal(batches[-1][0][pl].cpu(), ids8)
    mid8 = fc.to_dev(mi8)
    Lt = ids8.shape[1]

    def prefill_state(ids, mid, rows):
        """every layer on every row (forward()'s route), then the fixture rows' final hidden states (sequence order) and K / V entries"""
        feats, _ = model.encode_modal_inputs(mid, model.prefix_tokens, model.suffix_tokens)
        plan = model._plan(ids, None, None, mid, feats)
        st = model._prefill(plan, feats, fc.N_NEW, want_hidden=True, want_logits=False)
        torch.cuda.synchronize()
        L = int(plan.valid_lens[0])
        hid = model._rows_to_sequence(st["hidden"], st["out_map"], plan.B, plan.Lmax)[rows].clone()
        kc = st["kc"][:, rows, :, :L].clone()
        vc = st["vc"][:, rows, :, :L].clone()
        return hid, kc, vc, st["layout"].M

    hid8, kc8, vc8, M8 = prefill_state(ids8.cuda(), mid8, list(range(8)))
    res8, lg8 = model.generate(ids8.cuda(), modal_inputs=mid8, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True)
    ids8_new, lg8 = res8[:, Lt:].cpu(), lg8.float().cpu()
    rep = {"rows_per_batch": 48, "fixture_slots": place, "routed_rows_8": int(M8), "batches": []}
    for bi, (ids48, mid48) in enumerate(batches):
        hid, kc, vc, M48 = prefill_state(ids48, mid48, place[bi])
        same_h, same_k, same_v = torch.equal(hid, hid8), torch.equal(kc, kc8), torch.equal(vc, vc8)
        rep["batches"].append({"routed_rows": int(M48), "hidden_bitwise_equal": same_h, "k_cache_bitwise_equal": same_k, "v_cache_bitwise_equal": same_v})
        del hid, kc, vc
        assert M48 == 48 * 2793, M48
        assert same_h and same_k and same_v, rep
    del hid8, kc8, vc8
    torch.cuda.empty_cache()
    # (2) the shipped loop
    outs = list(model.generate_pipelined(batches, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True))
    ids_graph = [o[:, Lt:].cpu() for o in model.generate_pipelined(batches, max_new_tokens=fc.N_NEW, ignore_eos=True)]
    assert model.runtime_option("graph_active") == 1
    scale = lg8.abs().max()
    for bi, ((res, lg), idg) in enumerate(zip(outs, ids_graph)):
        got_ids, got_lg = res[:, Lt:].cpu()[place[bi]], lg.float().cpu()[place[bi]]
        assert torch.equal(idg[place[bi]], got_ids), "graph-replayed decode and one-launch-per-kernel decode disagree"
        e_all = ((got_lg - lg8).abs().amax(-1) / scale)                   # [8 rows, 17 steps]
        matched = [int((got_ids[r] != ids8_new[r]).nonzero()[0]) if (got_ids[r] != ids8_new[r]).any() else fc.N_NEW for r in range(8)]
        rep["batches"][bi].update(first_step_logit_diff_max=e_all[:, 0].max().item(), step_logit_diff_max=e_all.max().item(),
                                  step_logits_bitwise_equal=bool(torch.equal(got_lg, lg8)), tokens_equal_to_the_8_row_run=matched)
        assert matched == [fc.N_NEW] * 8, (bi, matched)
        assert torch.equal(got_lg, lg8), (bi, e_all.max().item())
    REPORT["benchmarked_shape_b48"] = rep
    _dump()
    print("b48", json.dumps(rep))


def _ulps(d, o):
    """|d - o| in units of the bf16 spacing at max(|d|, |o|, rms of the tensor) (as tests/test_fullwidth_parity_gpu.py)."""
    mag = torch.maximum(torch.maximum(d.abs(), o.abs()), o.pow(2).mean().sqrt())
    ulp = torch.exp2(torch.floor(torch.log2(mag)) - 7)
    return (d - o).abs() / ulp


# the layer's output is the residual stream - bf16(x1 + down(...)), x1 = bf16(x + o(...)) - so a flipped rounding upstream moves an output
# element by at most a few of ITS ulps.  Measured on MI355X (profiles/r05_parity.json): layer 0 (the stream is still as small as the layer's
# own contribution: rms 1.6 -> 2.8) max 4 ulps, 2.1 % of the elements over 1 ulp; layers 8 / 16 / 24 / 31 (rms 6.7 ... 12.9) max 3 ulps,
# 1.1e-3 / 1.9e-4 / 2.2e-4 / 2.2e-4 over 1 ulp.  Bounds ~2x.
LAYER_MAX_ULPS = 6.0
LAYER_FRAC_OVER_1 = {0: 4e-2}
LAYER_FRAC_OVER_1_DEEP = 2.5e-3


def test_every_probed_layer_is_exact_to_rounding_on_its_own_input(model32):
    """layers 1 .. 31 were never compared at ulp level - the depth tests bound 32 layers of accumulated bf16 noise
    (~3e-2 of the logit scale), under which a depth-dependent defect of a few ulp per layer would hide.  Here layer l in {0, 8, 16, 24, 31}
    of the 32-layer model is checked ALONE: forward(output_hidden_states=True) yields the hidden state that enters it and the one that
    leaves it on the device; the device-rounding restatement (oracle/device_path.forward_layer: same pre-merged weights, same storage
    points, CPU fp32 arithmetic) is handed the SAME input - the activations the layer really sees at that depth (the residual stream grows
    with depth), every routed adapter group, L = 2793 - and its output must agree with the device's to a few bf16 ulp on every element."""
    from oracle import device_path, pipeline
    model, meta, sd = model32
    name = "fulldepth_iav"
    ids, mi = fc.build_rows(name, row_seeds=[700])
    mid = fc.to_dev(mi)
    out = model.forward(input_ids=ids.cuda(), modal_inputs=mid, output_hidden_states=True)
    n = meta["num_hidden_layers"]
    assert len(out.hidden_states) == n + 1
    hs = [h.float().cpu() for h in out.hidden_states[:n]] + [out.raw_last_hidden_state.float().cpu()]      # input of layer l = hs[l], its output hs[l + 1]
    feats, _ = model.encode_modal_inputs(mid, model.prefix_tokens, model.suffix_tokens)
    sdf = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    od = pipeline.OracleModel.from_state_dict(sdf, meta, emulate="device", device_opts={"lazy": True})
    _, emb, _, mam = od.prepare(ids, fc.to_f32(mi), feats_blocks={m: f.float().cpu() for m, f in feats.items()})
    assert torch.equal(device_path.bf(emb.float()), hs[0]), "the spliced embeddings differ from the device's layer-0 input"
    dw = od.device_weights()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    rep = {}
    for l in PROBE_LAYERS:
        with torch.no_grad():
            o = device_path.forward_layer(dw, l, hs[l], mam)
        d = hs[l + 1]
        u = _ulps(d, o)
        rep[str(l)] = {"max_ulps": u.max().item(), "frac_differing": (u > 0).float().mean().item(), "frac_over_1ulp": (u > 1.0 + 1e-6).float().mean().item(),
                       "rms_in": hs[l].pow(2).mean().sqrt().item(), "rms_out": d.pow(2).mean().sqrt().item(),
                       "rel_err": ((d - o).abs().max() / o.abs().max()).item()}
        print("layer", l, rep[str(l)])
    REPORT["per_layer_teacher_forced"] = {"layers": rep, "bound_max_ulps": LAYER_MAX_ULPS, "bound_frac_over_1ulp_layer0": LAYER_FRAC_OVER_1[0],
                                          "bound_frac_over_1ulp_deeper": LAYER_FRAC_OVER_1_DEEP, "tokens": int(hs[0].shape[1])}
    _dump()
    for l, r_ in rep.items():
        assert r_["max_ulps"] <= LAYER_MAX_ULPS and r_["frac_over_1ulp"] <= LAYER_FRAC_OVER_1.get(int(l), LAYER_FRAC_OVER_1_DEEP), (l, r_)
