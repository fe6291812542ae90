"""Parity at the REAL widths against the CPU oracle: Vicuna-7B hidden 4096 / 32 x 128 / FFN 11008 / vocab 32000 / r = 128,
two decoder layers, real-size encoders, for BASELINE configs[1], configs[2] (3-way composed model, image + audio, video input absent)
and configs[3] (4 modalities, MCUB-4-shaped, spliced length 3337).  Every test goes through the C ABI (libmc_hip.so) on the GPU and
runs the oracle on the host cores of the same box.

Bars (reference: modelcompose/model/language_model/multimodal_llama.py:120-160, 210-342, 363-396;
eval/model_multimodal_qa_loader.py:94-108):
  * greedy ids equal on every row and every one of the 17 tokens (prefill token + 16 decode steps), against BOTH oracles, with no
    margin gate (row seeds are margin-screened on the CPU, tests/fullwidth_cases.py);
  * logits of all 17 steps vs the fp32 BRANCH-FORM oracle (the reference's arithmetic, its own fp32 encoders): within FP32_BOUND =
    2x the error measured on MI355X in round 2 (profiles/r02_parity.json);
  * logits vs the DEVICE-ROUNDING restatement (oracle/device_path.py: same bf16 storage points and pre-merged weights, fed the device's
    own encoder feature blocks): no further from it than NOISE_RATIO x the restatement's own distance from the fp32 oracle, i.e. the HIP
    path is no noisier than a CPU implementation with identical storage points.  A fixed 1e-3 is NOT attainable end to end: with bf16
    storage between kernels any difference in fp32 summation order (MFMA tiles vs a CPU GEMM) flips ~1e-3 of the roundings of the first
    GEMM output, and the flipped fraction f grows per stage as f' ~ sqrt(f)/2 to its fixed point 1/4 - two correct implementations
    decorrelate to the full bf16 noise level within one layer (measured here: q|k|v 0.04 % of elements differ, attention output 6 %,
    SwiGLU output 49 %).  What IS exact to rounding is each kernel on identical inputs:
  * first-layer stages, where both sides still see identical inputs: q|k|v of layer 0 (RMS factor + routed 256x256-tile GEMM against the
    norm-folded composed r=128 weights) every element within one bf16 ulp, its rotated q within four, <= 0.5 % of the elements differ at all;
    attention output (attn_prefill_kernel<128>): within three ulps, <= 15 % differ (online-softmax rounding of P relative to the running
    maximum instead of the final one).
"""
import json
import os
import time

import pytest
import torch

import fullwidth_cases as fc

pytestmark = pytest.mark.gpu

NOISE_RATIO = 2.0
FP32_BOUND = {"configs1_vision": 2.0e-2, "configs2_image_audio_video_absent": 2.0e-2, "configs3_mcub4": 2.0e-2}
# 2x the errors measured on MI355X in round 2 (bf16 encoders of 12-24 layers + projector vs the fp32 oracle, of the feature scale):
# vision 1.2e-2, audio 7.5e-3, video 1.8e-2, point 1.6e-2 (profiles/r02_parity.json)
ENC_BOUND = {"vision": 2.5e-2, "audio": 1.5e-2, "video": 3.6e-2, "point": 3.2e-2}
REPORT = {}


def rel(a, b):
    return ((a.float().cpu() - b.float().cpu()).abs().max() / b.float().abs().max()).item()


@pytest.fixture(scope="module", params=list(fc.CASES))
def case(request):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd.model.builder import build_from_state_dict
    from oracle import pipeline
    name = request.param
    t0 = time.time()
    c = fc.CASES[name]
    meta, sd, ids, mi = fc.build_case(name, c["row_seeds"] + c.get("extra_rows", []))
    model = build_from_state_dict(meta, sd)
    if "point" in mi:
        model.model.modal_encoders["point"].fps_start = torch.zeros(ids.shape[0], dtype=torch.long)
    mid = fc.to_dev(mi)
    feats, _ = model.encode_modal_inputs(mid, model.prefix_tokens, model.suffix_tokens)
    res, lg = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True)
    res_graph = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True)        # shipped path: hipGraph replay
    graph_active = model.runtime_option("graph_active")
    got = dict(ids=res[:, ids.shape[1]:].cpu(), ids_graph=res_graph[:, ids.shape[1]:].cpu(), logits=lg.float().cpu(),
               feats={m: f.float().cpu() for m, f in feats.items()}, graph_active=graph_active, names=list(model.modal_names))
    del model, feats
    torch.cuda.empty_cache()
    sdf = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    del sd
    t1 = time.time()
    with torch.no_grad():
        od = pipeline.OracleModel.from_state_dict(sdf, meta, emulate="device")
        ids_d, lg_d = od.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True, feats_blocks=got["feats"])
        od._dw = None
        o32 = pipeline.OracleModel.from_state_dict(sdf, meta)
        fns = {m: (lambda x, m=m: o32.encode_modal(m, x)) for m in o32.modals}
        from oracle import splice
        f32_feats, _ = splice.encode_modal_inputs(fc.to_f32(mi), o32.modals, fns, o32.prefix, o32.suffix, skip_absent=True)
        ids_r, lg_r = o32.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True, feats_blocks=f32_feats)
    t2 = time.time()
    ref = dict(ids_device=ids_d, logits_device=lg_d, ids_fp32=ids_r, logits_fp32=lg_r, feats_fp32=f32_feats)
    spliced = int(ids.shape[1] - len(got["feats"]) + sum(f.shape[1] for f in got["feats"].values()))
    REPORT[name] = {"hip_s": round(t1 - t0, 1), "oracle_s": round(t2 - t1, 1), "spliced_length": spliced, "batch": int(ids.shape[0])}
    # rows [0, ns) are the screened rows (exact tests); the rest are unscreened (property test)
    ns = len(c["row_seeds"])
    extra_got = {k: (v[ns:] if torch.is_tensor(v) else v) for k, v in got.items()}
    extra_ref = {k: (v[ns:] if torch.is_tensor(v) else v) for k, v in ref.items()}
    for d_ in (got, ref):
        for k in list(d_):
            if torch.is_tensor(d_[k]):
                d_[k] = d_[k][:ns]
    got["extra"], ref["extra"] = extra_got, extra_ref
    got["feats_all"], ref["feats_all"] = got["feats"], ref["feats_fp32"]
    yield name, got, ref
    out = os.path.join(fc.ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(REPORT, open(os.path.join(out, "fullwidth_parity.json"), "w"), indent=1)
    except OSError:
        pass


def test_greedy_ids_equal_on_every_row_and_step(case):
    name, got, ref = case
    n_cmp = int(ref["ids_device"].numel())
    REPORT[name].update(ids_compared=n_cmp, ids_equal_device_oracle=bool(torch.equal(got["ids"], ref["ids_device"])),
                        ids_equal_fp32_oracle=bool(torch.equal(got["ids"], ref["ids_fp32"])),
                        min_margin_device_oracle=fc.margins(ref["logits_device"]).min().item(),
                        min_margin_fp32_oracle=fc.margins(ref["logits_fp32"]).min().item())
    assert n_cmp == len(fc.CASES[name]["row_seeds"]) * fc.N_NEW and got["ids"].shape == ref["ids_fp32"].shape
    assert torch.equal(got["ids"], ref["ids_fp32"]), (got["ids"], ref["ids_fp32"])            # every row, every step, no margin gate
    assert torch.equal(got["ids"], ref["ids_device"]), (got["ids"], ref["ids_device"])


def test_logits_no_noisier_than_the_device_rounding_restatement(case):
    name, got, ref = case
    err_dev_emu = rel(got["logits"], ref["logits_device"])
    err_emu_fp32 = rel(ref["logits_device"], ref["logits_fp32"])
    REPORT[name].update(err_vs_device_oracle=err_dev_emu, device_oracle_vs_fp32_oracle=err_emu_fp32)
    assert got["logits"].shape == ref["logits_device"].shape and got["logits"].shape[1] == fc.N_NEW
    assert err_dev_emu <= NOISE_RATIO * err_emu_fp32, (err_dev_emu, err_emu_fp32)
    assert err_dev_emu <= FP32_BOUND[name]


def test_unscreened_rows_agree_up_to_a_near_tie(case):
    """Rows nobody selected: greedy decoding is a discontinuous function of the logits, so a row may leave the oracle's path - but only at
    a step whose oracle top-2 margin is within the bf16 noise, and only to a token whose oracle logit lies within that band of the oracle's
    best.  Up to and including that step the
    logits are comparable and must be within the bound."""
    name, got, ref = case
    g, r = got["extra"], ref["extra"]
    n_rows = g["ids"].shape[0]
    if n_rows == 0:
        pytest.skip("no unscreened rows for this case (oracle cost)")
    NEAR_TIE = 1.5e-2            # of the logit scale: ~1.5x the measured max logit error, i.e. several sigma of the error of a logit difference
    full, rep = 0, []
    for key in ("fp32", "device"):
        ids_o, lg_o = r["ids_" + key], r["logits_" + key]
        scale = lg_o.abs().max()
        for b in range(n_rows):
            neq = (g["ids"][b] != ids_o[b]).nonzero()
            t = int(neq[0]) if len(neq) else fc.N_NEW
            # logits of the steps that saw identical inputs (all steps up to and including the first differing one)
            upto = min(t + 1, fc.N_NEW)
            err = ((g["logits"][b, :upto] - lg_o[b, :upto]).abs().max() / scale).item()
            assert err <= FP32_BOUND[name], (key, b, err)
            if t == fc.N_NEW:
                full += 1
                continue
            top2 = lg_o[b, t].topk(2)
            margin = ((top2.values[0] - top2.values[1]) / scale).item()
            # the token the device chose instead: its ORACLE logit must itself be within the near-tie band of the oracle's best (with
            # random-weight models three candidates inside the band occur; the device may take any of them, not only the runner-up)
            chosen_gap = ((top2.values[0] - lg_o[b, t, int(g["ids"][b, t])]) / scale).item()
            rep.append({"oracle": key, "row": b, "step": t, "margin": margin, "chosen_gap": chosen_gap})
            assert margin < NEAR_TIE, (key, b, t, margin)
            assert chosen_gap < NEAR_TIE, (key, b, t, chosen_gap)
    REPORT[name].update(unscreened_rows=n_rows, unscreened_fully_equal=full, unscreened_total=2 * n_rows, unscreened_near_tie_departures=rep)
    # How MANY rows meet a near-tie within 17 steps is a property of the random-weight model's margin distribution, not of the kernels
    # (9 of 12 pairs stayed on the oracle's path with one build of this round, 5 of 12 with the next, at an unchanged logit error of
    # 6.6e-3): it is reported, not asserted.  What is asserted above: no departure outside the band, logits within the bound up to it.


def test_graph_replayed_decode_gives_the_same_ids(case):
    name, got, ref = case
    assert got["graph_active"] == 1
    assert torch.equal(got["ids_graph"], got["ids"]) and torch.equal(got["extra"]["ids_graph"], got["extra"]["ids"])


def test_against_the_fp32_branch_form_oracle(case):
    name, got, ref = case
    err = rel(got["logits"], ref["logits_fp32"])
    REPORT[name].update(err_vs_fp32_oracle=err)
    assert err <= FP32_BOUND[name], f"{name}: {err:.2e}"


def test_encoder_feature_blocks(case):
    name, got, ref = case
    assert set(got["feats"]) == set(fc.CASES[name]["inputs"])              # absent modalities are not encoded (configs[2]: no video block)
    errs = {}
    for m, f in got["feats_all"].items():
        assert f.shape == ref["feats_all"][m].shape, m
        errs[m] = rel(f, ref["feats_all"][m])
    REPORT[name]["encoder_err"] = errs
    for m, e in errs.items():
        assert e <= ENC_BOUND[m], (m, e)


def _ulps(d, o):
    """|d - o| in units of the bf16 spacing at max(|d|, |o|, rms of the tensor): elements much smaller than the tensor's typical magnitude
    are sums with cancellation, whose fp32 summation-order error is set by the size of the terms, not of the result."""
    mag = torch.maximum(torch.maximum(d.abs(), o.abs()), o.pow(2).mean().sqrt())
    ulp = torch.exp2(torch.floor(torch.log2(mag)) - 7)
    return (d - o).abs() / ulp


@pytest.mark.parametrize("name", ["configs1_vision", "configs2_image_audio_video_absent"])
def test_first_layer_stages_are_exact_to_rounding(name):
    """One decoder layer: the runtime's workspace still holds layer 0's q|k|v, rotated q and attention output after the prefill; the
    restatement computes them from the same embeddings (see the module docstring for why only the first stages can be held to one ulp)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import numpy as np
    from modelcompose_amd.model.builder import build_from_state_dict
    from oracle import device_path, pipeline
    meta, sd, ids, mi = fc.build_case(name)
    meta["num_hidden_layers"] = 1
    sd = {k: v for k, v in sd.items() if not k.startswith("model.layers.1.")}
    model = build_from_state_dict(meta, sd)
    model.last_layer_tail = False            # this one-layer model's only layer must run for every row: its stages are what is compared
    mid = fc.to_dev(mi)
    feats, _ = model.encode_modal_inputs(mid, model.prefix_tokens, model.suffix_tokens)
    plan = model._plan(ids.cuda(), None, None, mid, feats)
    from modelcompose_amd import _lib
    cfg = model.config
    Hd, H, D = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim
    al = lambda v: (v + 255) // 256 * 256

    def stages():
        # the shipped route rotates q / k in the q|k|v projection's epilogue and never stores the un-rotated projection (the separate GEMM ->
        # mc_rope_kv_bf16 sequence is bit-identical: tests/test_ops_gpu.py::test_qkv_projection_with_rope_scatter_epilogue): the rotated
        # queries and the attention output are the stages compared here
        st_ = model._prefill(plan, feats, 0, want_hidden=False, want_logits=True)
        torch.cuda.synchronize()
        lay_ = st_["layout"]
        ws, off, v = st_["ws"], 0, {}
        for tag, rows, cols in (("qkv", lay_.M, 3 * H * D), ("q_rot", plan.B * plan.Lmax, H * D), ("attn", lay_.M, Hd)):     # carve() of csrc/llm_runtime.cpp
            v[tag] = ws[off:off + rows * cols * 2].view(torch.bfloat16).view(rows, cols).float().cpu()
            off += al(rows * cols * 2)
        return st_, v
    st, views = stages()
    lay = st["layout"]
    M, B, Lq = lay.M, plan.B, plan.Lmax
    sdf = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    od = pipeline.OracleModel.from_state_dict(sdf, meta, emulate="device")
    fb = {m: f.float().cpu() for m, f in feats.items()}
    _, emb, _, mam = od.prepare(ids, fc.to_f32(mi), feats_blocks=fb)
    tr = {}
    with torch.no_grad():
        device_path.forward(od.device_weights(), device_path.bf(emb.float()), mam, last_only=True, trace=tr)
    seq_of_row = torch.from_numpy(lay.order_b.astype(np.int64) * Lq + lay.order_t.astype(np.int64))
    rep = {}
    # q|k|v: one rounding away at most.  Rotated q = a*cos - b*sin of two values that may each be one ulp off, measured in ulps of the
    # (possibly cancelled) result: up to 3 seen on 1.6e-6 of the elements; attention output: the same through the P.V sum (2 seen)
    for tag, max_frac, max_ulps in (("q_rot", 0.005, 4.0), ("attn", 0.15, 3.0)):
        d = views[tag]
        o = tr["0." + tag]
        o = o[seq_of_row] if tag != "q_rot" else o
        u = _ulps(d, o)
        rep[tag] = {"max_ulps": u.max().item(), "frac_differing": (u > 0).float().mean().item(),
                    "frac_over_1ulp": (u > 1.0 + 1e-6).float().mean().item(), "bound_ulps": max_ulps, "bound_frac": max_frac}
    REPORT.setdefault("first_layer_stages", {})[name] = rep
    print("first-layer stages", name, rep)
    for tag, r_ in rep.items():
        assert r_["max_ulps"] <= r_["bound_ulps"] + 1e-6 and r_["frac_differing"] <= r_["bound_frac"], (tag, rep)
    out = os.path.join(fc.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(REPORT, open(os.path.join(out, "fullwidth_parity.json"), "w"), indent=1)


def test_small_delta_composition_against_the_branch_form():
    """VERDICT r1 #2 / r2 #2 / SURVEY §7 "hard parts": W' = bf16(W + sum s B A) drops a delta smaller than half a bf16 step of W.  LoRA B
    scaled so that |dW| / |W| ~ 2^-10 elementwise (the adversarial regime; trained deltas are ~2^-4).  The whole delta then moves the
    logits by less than the path's bf16 noise, so "logits within the bound" cannot tell a kept delta from a dropped one.  What can:
    the DIFFERENCE the delta makes.  dH = HIP(with delta) - HIP(B = 0) and dO = oracle(with delta) - oracle(B = 0) on the first-step
    logits (fp32 branch-form oracle, multimodal_llama.py:130-149, which keeps the delta exactly); the projection c = <dH, dO> / <dO, dO>
    averages the bf16 noise of the two device runs (uncorrelated with dO) over 64 000 logits: c ~ 0 if the composition lost the delta,
    ~1 if it kept all of it, ~0.7 for RNE rounding of on-grid base weights at this ratio (tests/test_ops_gpu.py::
    test_compose_small_delta_retention_is_the_rne_value pins that value at the op level; round 3 shipped that).  Round 4: finalize()
    detects the loss (retention < 0.9) and composes those adapters with UNBIASED rounding instead: c must be ~1."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import warnings
    from modelcompose_amd.model.builder import build_from_state_dict
    from oracle import pipeline
    name = "configs1_vision"
    meta, sd, ids, mi = fc.build_case(name, lora_b_std=0.01 / 64)
    w = sd["model.layers.0.self_attn.q_proj.weight"].float()
    dw = 2.0 * sd["model.layers.0.self_attn.q_proj.lora_B.default.weight"].float() @ sd["model.layers.0.self_attn.q_proj.lora_A.default.weight"].float()
    ratio = (dw.abs().mean() / w.abs().mean()).item()
    assert 2 ** -11.5 < ratio < 2 ** -8.5, ratio
    mid = fc.to_dev(mi)

    sdf = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    sd_zero = {k: (torch.zeros_like(v) if ".lora_B." in k else v) for k, v in sd.items()}
    forced = [None]

    def hip_logits(state):
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            model = build_from_state_dict(meta, state)
        feats, _ = model.encode_modal_inputs(mid, model.prefix_tokens, model.suffix_tokens)
        res, lg = model.generate(ids.cuda(), modal_inputs=mid, max_new_tokens=fc.N_NEW, ignore_eos=True, return_step_logits=True,
                                 **({"forced_ids": forced[0]} if forced[0] is not None else {}))
        out = (res[:, ids.shape[1]:].cpu(), lg.float().cpu(), {m: f.float().cpu() for m, f in feats.items()},
               {"final": dict(model.delta_retention), "rne": dict(getattr(model, "delta_retention_rne", {})), "dithered": list(getattr(model, "delta_dithered", []))},
               [str(w_.message) for w_ in wlist if issubclass(w_.category, RuntimeWarning)])
        del model
        torch.cuda.empty_cache()
        return out
    got_ids, lg_free, fb, retention, warned = hip_logits(sd)            # free running: the ids comparison
    with torch.no_grad():
        o32 = pipeline.OracleModel.from_state_dict(sdf, meta)
        ids_r, lg_r = o32.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True, feats_blocks=fb)
        # the same model with the LoRA terms removed, TEACHER-FORCED on the same history: all 17 steps are comparable, 17 x the samples of
        # the first step alone (the dither's zero-mean weight noise does not cancel between the two device runs: more samples, not a
        # looser bound, is what keeps the estimate of c sharp)
        sd0 = {k: (torch.zeros_like(v) if ".lora_B." in k else v) for k, v in sdf.items()}
        o0 = pipeline.OracleModel.from_state_dict(sd0, meta)
        _, lg_0 = o0.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True, feats_blocks=fb, forced_ids=ids_r)
    forced[0] = ids_r[:, :fc.N_NEW - 1]
    _, lg, fb1, _, _ = hip_logits(sd)
    _, lg_h0, fb0, retention0, _ = hip_logits(sd_zero)
    assert all(torch.equal(fb[m], fb0[m]) and torch.equal(fb[m], fb1[m]) for m in fb)      # the encoders do not depend on the LoRA terms
    dH = (lg - lg_h0).double()
    dO = (lg_r - lg_0).double()
    c = ((dH * dO).sum() / (dO * dO).sum()).item()
    c_steps = ((dH * dO).sum((0, 2)) / (dO * dO).sum((0, 2))).tolist()
    resid = ((dH - c * dO).norm() / dO.norm()).item()             # what is left after the projection: the two device runs' bf16 noise
    err = rel(lg[:, :1], lg_r[:, :1])
    effect = rel(lg_0[:, :1], lg_r[:, :1])
    agree = int((got_ids == ids_r).sum())
    REPORT["small_delta"] = {"dw_over_w": ratio, "prefill_err_vs_branch_form": err, "delta_effect_on_logits": effect,
                             "delta_projection_hip_on_oracle": c, "delta_projection_per_step": c_steps, "orthogonal_residual_over_effect": resid,
                             "compose_retention_per_adapter": retention, "warned": warned,
                             "ids_agree": agree, "ids_total": int(ids_r.numel()), "min_margin": fc.margins(lg_r).min().item()}
    out = os.path.join(fc.ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(REPORT, open(os.path.join(out, "fullwidth_parity.json"), "w"), indent=1)
    print("small delta:", REPORT["small_delta"])
    # Round 4: the delta is KEPT.  Round-to-nearest alone retains 0.6-0.8 of it at this ratio (`rne`: what the first compose
    # pass measured; round 3 shipped that and warned); finalize() now re-composes such adapters with the unbiased rounding of
    # mc_compose_weight_dither_bf16, so (1) the projection of the delta's effect on the oracle's is ~1, (2) the compose kernel's own
    # retention statistic is ~1, (3) no warning is raised.
    assert 0.95 <= c <= 1.05, f"delta effect projection {c:.3f} (per step {c_steps}): the composed weights lost (or inflated) the small delta"
    assert retention["rne"] and all(0.5 < v < 0.9 for v in retention["rne"].values()), retention
    assert sorted(retention["dithered"]) == sorted(retention["rne"]) and all(0.99 < v < 1.01 for v in retention["final"].values()), retention
    assert not warned, warned
    assert not retention0["final"] and not retention0["dithered"]   # B = 0: no delta, nothing to retain, no statistic
    # (3) and the composed path stays within the same bound as with ordinary deltas
    assert err <= FP32_BOUND[name], f"pre-merged weights: {err:.2e} of the logit scale (delta effect {effect:.2e})"
