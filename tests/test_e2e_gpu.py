"""End-to-end parity on the GPU: the HIP path (through the C ABI) against the golden fixtures produced by the
reference and against the CPU oracle on seeded tiny configurations."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g4_model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd.model.builder import build_from_state_dict
    a, meta, sd = load_golden("g4_e2e_vision")
    return build_from_state_dict(meta, sd), a, meta, sd


def rel_err(got, ref):
    return (got.float().cpu() - ref.float()).abs().max().item() / ref.float().abs().max().item()


def within(tag, got, ref, bound):
    """max |got - ref| / max |ref| < bound; the measured value is printed (pytest -s) so that bounds can be kept at ~2x of it."""
    e = rel_err(got, ref)
    print(f"[err] {tag}: {e:.3e} (bound {bound:.1e})")
    assert e < bound, (tag, e, bound)


def test_clip_tower_matches_golden():
    from modelcompose_amd.model.clip import ClipVisionConfig, HipClipVisionTower
    a, meta, sd = load_golden("g5_clip")
    tower = HipClipVisionTower(None, None, delay_load=True, config=ClipVisionConfig(**meta))
    tower.load_state_dict(sd)
    f = tower(a["pixels"].cuda())
    # bf16 storage through 2 transformer layers vs the fp32 reference: 2^-6 of the feature scale
    within("1 clip features", f, a["features"], 2 ** -6)
    tower.select_layer, tower.select_feature = -1, "cls_patch"
    within("2 clip last layer cls_patch", tower(a["pixels"].cuda()), a["features_last_cls"], 2 ** -6)


def test_g4_prefill_logits_and_greedy_ids(g4_model):
    model, a, meta, sd = g4_model
    ids = a["input_ids"].cuda()
    px = a["pixels"].cuda()
    out = model.forward(input_ids=ids, modal_inputs={"vision": px})
    assert out.logits.shape == a["logits_prefill"].shape
    # tolerances: 2x the error measured on MI355X in round 2 (bf16 storage vs the fp32 reference over 2 layers), of the logit scale
    within("3 g4 prefill logits", out.logits, a["logits_prefill"], 1.2e-2)
    n_new = a["gen_ids"].shape[1]
    res, step_logits = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=n_new, ignore_eos=True,
                                      return_step_logits=True)
    assert res.shape == (ids.shape[0], ids.shape[1] + n_new)
    assert torch.equal(res[:, :ids.shape[1]].cpu(), a["input_ids"])
    within("4 g4 step logits", step_logits, a["step_logits"], 1e-2)          # measured 5.0e-3
    # greedy ids: equal to the reference's on every row and every step (the fixture's margins: see the assertion message on failure)
    got = res[:, ids.shape[1]:].cpu()
    assert got.shape == a["gen_ids"].shape and got.numel() == ids.shape[0] * n_new
    assert torch.equal(got, a["gen_ids"]), (got, a["gen_ids"], _margins(a["step_logits"]))


def test_output_hidden_states_and_attentions_match_the_reference(g4_model):
    """forward(output_hidden_states=True, output_attentions=True) (multimodal_llama.py:561-604, :295-312, :676-745) against the reference's
    own outputs on g4's model (tests/golden/g18_hidden_attn.npz, written by `python -m oracle.gen_golden g18`): the tuple of hidden states -
    the input of every decoder layer, then the final norm of the last layer's output - and every layer's attention probabilities."""
    model, a, meta, sd = g4_model
    g18, _, _ = load_golden("g18_hidden_attn")
    assert torch.equal(g18["input_ids"], a["input_ids"]) and torch.equal(g18["logits_prefill"], a["logits_prefill"])      # same model, same inputs
    ids, px = a["input_ids"].cuda(), a["pixels"].cuda()
    plain = model.forward(input_ids=ids, modal_inputs={"vision": px})
    out = model.forward(input_ids=ids, modal_inputs={"vision": px}, output_hidden_states=True, output_attentions=True)
    assert plain.hidden_states is None and plain.attentions is None
    assert torch.equal(out.logits, plain.logits), "capturing must not change what is computed"
    n_layers = meta["num_hidden_layers"]
    assert isinstance(out.hidden_states, tuple) and len(out.hidden_states) == n_layers + 1 and len(out.attentions) == n_layers
    ref_h, ref_a = g18["hidden_states"], g18["attentions"]
    for l in range(n_layers + 1):
        assert out.hidden_states[l].shape == ref_h[l].shape
        # bf16 storage vs the fp32 reference (entry 0 = the spliced embeddings: text rows exact to one rounding, the image block as far
        # from fp32 as the bf16 CLIP tower + projector: measured 5.6e-3)
        within(f"11 hidden_states[{l}]", out.hidden_states[l], ref_h[l], 1.2e-2)
    for l in range(n_layers):
        assert out.attentions[l].shape == ref_a[l].shape
        got = out.attentions[l].float().cpu()
        assert (got - ref_a[l]).abs().max().item() < 1.5e-2, (l, (got - ref_a[l]).abs().max().item())       # probabilities: absolute
        assert (got[ref_a[l] == 0] == 0).all()                                                           # the causal zeros are exact zeros
        assert (got.sum(-1) - 1).abs().max().item() < 2e-2
    # the un-normed last layer output rides along (used by the per-layer parity test at depth)
    assert out.raw_last_hidden_state.shape == out.hidden_states[-1].shape


def _margins(ref_logits):
    top2 = ref_logits.topk(2, dim=-1).values
    return ((top2[..., 0] - top2[..., 1]) / ref_logits.abs().max()).tolist()


def test_g4_against_the_device_rounding_oracle_ids_exact(g4_model):
    """Greedy ids equal on every row and step against the oracle run with the device path's rounding points (bf16 storage, pre-merged
    norm-folded weights: oracle/device_path.py), fed the device's own feature block; logits no further from it than from the fp32 reference."""
    from oracle import pipeline
    model, a, meta, sd = g4_model
    sd16 = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() else v) for k, v in sd.items()}
    om = pipeline.OracleModel.from_state_dict(sd16, meta, emulate="device")
    n_new = 8
    mi = {"vision": a["pixels"].cuda()}
    feats, _ = model.encode_modal_inputs(mi, model.prefix_tokens, model.suffix_tokens)
    ids_o, lg_o = om.generate(a["input_ids"], {"vision": a["pixels"]}, max_new_tokens=n_new, ignore_eos=True, return_logits=True,
                              feats_blocks={m: f.float().cpu() for m, f in feats.items()})
    res, lg = model.generate(a["input_ids"].cuda(), modal_inputs=mi, max_new_tokens=n_new, ignore_eos=True, return_step_logits=True)
    within("5 g4 vs device-rounding oracle", lg, lg_o, 1e-2)
    got = res[:, a["input_ids"].shape[1]:].cpu()
    assert got.numel() == a["input_ids"].shape[0] * n_new and torch.equal(got, ids_o), (got, ids_o, _margins(lg_o))


def test_generate_eos_padding_and_reference_shape(g4_model):
    model, a, meta, sd = g4_model
    ids = a["input_ids"].cuda()
    res = model.generate(ids, modal_inputs={"vision": a["pixels"].cuda()}, max_new_tokens=6, do_sample=False, temperature=0,
                         num_beams=1, use_cache=True)
    assert res.dtype == torch.int64 and res.shape[0] == ids.shape[0] and res.shape[1] <= ids.shape[1] + 6
    with pytest.raises(NotImplementedError):                  # beam search is greedy-scored only (round 5: num_beams > 1 itself is built)
        model.generate(ids, modal_inputs={"vision": a["pixels"].cuda()}, num_beams=2, do_sample=True, temperature=0.7)
    # the loader's default (--temperature 0.2 -> do_sample=True, model_multimodal_qa_loader.py:96-99) runs on the sampled path
    res = model.generate(ids, modal_inputs={"vision": a["pixels"].cuda()}, max_new_tokens=6, do_sample=True, temperature=0.2, top_p=None)
    assert res.dtype == torch.int64 and res.shape[0] == ids.shape[0] and res.shape[1] <= ids.shape[1] + 6


def test_splice_api_matches_golden_g3():
    """prepare_inputs_labels_for_multimodal through the HIP copy kernels: masks/labels bit-exact, embeddings exact in bf16."""
    from modelcompose_amd.model.config import MultimodalConfig
    from modelcompose_amd.model.multimodal_llama import MultimodalLlamaForCausalLM
    a, _, _ = load_golden("g3_splice")
    H = a["embed_tokens"].shape[1]
    cfg = MultimodalConfig(vocab_size=a["embed_tokens"].shape[0], hidden_size=H, num_attention_heads=2, num_hidden_layers=1,
                           intermediate_size=64, mm_vision_encoder="x", mm_audio_encoder="x", mm_video_encoder="x",
                           lora_strategy="modal+language")
    m = MultimodalLlamaForCausalLM(cfg)
    m.model.embed_tokens = a["embed_tokens"].cuda().to(torch.bfloat16)

    class Enc:
        def __call__(self, x=None, **kw):
            return kw["audio_inputs"] if x is None else x

    for k in ("audio", "vision", "video"):
        m.model.modal_encoders[k] = Enc()
        m.model.modal_projectors[k] = lambda t: t
    pre = {k.split("::")[1]: v.cuda().to(torch.bfloat16).reshape(-1, H) for k, v in a.items() if k.startswith("prefix::")}
    suf = {k.split("::")[1]: v.cuda().to(torch.bfloat16).reshape(-1, H) for k, v in a.items() if k.startswith("suffix::")}
    for tag, keys, use_ps in (("eq", ["vision", "audio"], True), ("ragged", ["vision", "video"], True), ("edge", ["vision", "video"], False)):
        mi = {}
        for k in keys:
            t = a[f"{tag}::modal::{k}"].cuda().to(torch.bfloat16)
            mi[k] = {"audio_inputs": t} if k == "audio" else t
        labels = a.get(f"{tag}::labels_in")
        _, am, _, emb, lab, mam = m.prepare_inputs_labels_for_multimodal(
            a[f"{tag}::input_ids"], a[f"{tag}::attention_mask_in"], None, labels, mi, pre if use_ps else None, suf if use_ps else None)
        assert torch.equal(am, a[f"{tag}::attention_mask"])
        if labels is not None:
            assert torch.equal(lab, a[f"{tag}::labels"])
        exp = {k.split("::")[-1]: v for k, v in a.items() if k.startswith(f"{tag}::mask::")}
        assert set(mam) == set(exp)
        for k in exp:
            assert torch.equal(mam[k], exp[k]), (tag, k)
        assert torch.equal(emb.cpu(), a[f"{tag}::embeds"].to(torch.bfloat16))


def test_g8_four_modality_composed_model_matches_reference():
    """BASELINE configs 3/4 in miniature: CLIP + BEATs/Q-Former + LanguageBind-Video + PointBERT feature blocks spliced in two
    different orders, 4-way online-merge-reset coefficients, routed prefill, device-resident greedy decode — against the
    reference's own outputs (tests/golden/g8_e2e_4modal.npz)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd.model.builder import build_from_state_dict
    a, meta, sd = load_golden("g8_e2e_4modal")
    model = build_from_state_dict(meta, sd)
    assert model.modal_names == ["default", "audio", "vision", "video", "point"]
    mi = {"vision": a["pixels"].cuda(), "audio": {"audio_inputs": a["fbank"].cuda(), "audio_padding_mask": a["padding_mask"].cuda()},
          "video": a["video"].cuda(), "point": a["points"].cuda()}
    feats, masks = model.encode_modal_inputs(mi, model.prefix_tokens, model.suffix_tokens)
    for m in ("vision", "audio", "video", "point"):
        assert feats[m].shape == a[f"feat_{m}"].shape
        # bf16 encoders (2-3 layers) + projector vs the fp32 reference; point: bf16 rounding of the coordinates on top
        within(f"6 feats[{m}]", feats[m], a[f"feat_{m}"], 2e-2)       # measured: audio 1.0e-2, video 8.9e-3, point 9.2e-3
    # round 4: the towers run on side streams by default (each tower's launches fill the others' partly filled rounds of tiles); one stream,
    # and a tower's batch split over streams, compute the same bits - every launch computes what it computes alone
    assert getattr(model, "encode_streams", True)
    for streams, split in ((False, 1), (True, 2)):
        model.encode_streams, model.encode_split = streams, split
        try:
            f2, _ = model.encode_modal_inputs(mi, model.prefix_tokens, model.suffix_tokens)
        finally:
            del model.encode_streams, model.encode_split
        for m in feats:
            assert torch.equal(f2[m], feats[m]), (m, streams, split)
    ids = a["input_ids"].cuda()
    out = model.forward(input_ids=ids, modal_inputs=mi)
    assert out.logits.shape == a["logits_prefill"].shape
    within("7 g8 prefill logits", out.logits, a["logits_prefill"], 2.5e-2)      # measured 1.2e-2
    n_new = a["gen_ids"].shape[1]
    res, step_logits = model.generate(ids, modal_inputs=mi, max_new_tokens=n_new, ignore_eos=True, return_step_logits=True)
    within("8 g8 step logits", step_logits, a["step_logits"], 1.5e-2)           # measured 7.4e-3
    got = res[:, ids.shape[1]:].cpu()
    assert got.numel() == ids.shape[0] * n_new
    assert torch.equal(got, a["gen_ids"]), (got, a["gen_ids"], _margins(a["step_logits"]))


@pytest.mark.parametrize("k,lp", [(2, 1.0), (3, 1.0), (3, 0.0)])
def test_beam_search_equals_the_restated_transformers_loop(g4_model, k, lp):
    """generate(num_beams=k) (eval/model_multimodal_qa_loader.py:94-102 forwards --num_beams) against oracle/beam.py - transformers 4.31's
    beam_search + BeamSearchScorer restated and pinned against the installed transformers (tests/test_beam_cpu.py) - driven by the fp32 oracle
    model on the same prompts: an image prompt (the reference itself fails on those: it expands input_ids k-fold but not modal_inputs,
    multimodal_arch.py:343-346) and a text-only batch.  Sequences must be equal unless the oracle's own candidate ranking had a near-tie."""
    from oracle import beam, pipeline
    model, a, meta, sd = g4_model
    om = pipeline.OracleModel.from_state_dict(sd, meta)
    n_new = 6
    eos, pad = meta.get("eos_token_id", 2), meta.get("pad_token_id", 0)

    def pad_to(x, n):
        return torch.cat([x, torch.full((x.shape[0], n - x.shape[1]), pad, dtype=x.dtype)], 1) if x.shape[1] < n else x

    def run(ids, mi_dev, mi_cpu):
        model._beam_trace = []
        try:
            got = model.generate(ids.cuda(), modal_inputs=mi_dev, num_beams=k, max_new_tokens=n_new, length_penalty=lp).cpu()
            trace = model._beam_trace
        finally:
            model._beam_trace = None
        rep = lambda d, rows: {m: v.repeat_interleave(rows.shape[0] // ids.shape[0], 0) for m, v in d.items()}

        def oracle_logits(noise=0.0, seed=0):
            gen = torch.Generator().manual_seed(seed)

            def f(rows):
                with torch.no_grad():
                    lg, _, _ = om.prefill(rows, rep(mi_cpu, rows), last_only=True)
                l = lg[:, -1]
                return l + noise * l.abs().max() * torch.randn(l.shape, generator=gen) if noise else l
            return f
        bs = lambda f: beam.beam_search(f, ids, k, n_new, eos, pad, length_penalty=lp)
        W = ids.shape[1] + n_new
        # (1) the beam bookkeeping in isolation: the restated loop, fed the very logits the device scored each of its beam rows with (looked
        # up by the row's ids - a row the device never scored is a KeyError), must walk the same beams and return the same sequences
        table = {}
        for rows, lg in trace:
            for r_ in range(rows.shape[0]):
                table[tuple(rows[r_].tolist())] = lg[r_]
        want_dev = bs(lambda rows: torch.stack([table[tuple(r_.tolist())] for r_ in rows]))
        assert torch.equal(pad_to(got, W), pad_to(want_dev, W)), (got, want_dev)
        # (2) the KV rows follow their beams: every step's logits (cached decode over replicated / gathered cache rows) against a fresh
        # prefill of the same rows through forward() - two device paths, each exact to rounding
        for rows, lg in trace:
            fresh = model.forward(input_ids=rows.cuda(), modal_inputs=rep(mi_dev, rows)).logits[:, -1].float().cpu()
            assert float((lg - fresh).abs().max() / fresh.abs().max()) < 1.2e-2
        # (3) against the fp32 oracle model: every row is the oracle's hypothesis, or one the oracle itself reaches when its logits are
        # perturbed by a third of the fixture's logit tolerance (sigma 4e-3 of the logit scale; tolerance 1.2e-2, test_g4_*) - a hypothesis
        # decided by an EOS candidate at rank k - 1 vs k is a near-tie that bf16 may resolve the other way
        want = pad_to(bs(oracle_logits()), W)
        reach = [pad_to(bs(oracle_logits(4e-3, sd_)), W) for sd_ in range(8)]
        g_ = pad_to(got, W)
        exact = torch.tensor([torch.equal(g_[b], want[b]) for b in range(ids.shape[0])])
        for b in range(ids.shape[0]):
            assert exact[b] or any(torch.equal(g_[b], nz[b]) for nz in reach), (b, g_[b], want[b])
        return got, want, exact
    got, want, ex_img = run(a["input_ids"], {"vision": a["pixels"].cuda()}, {"vision": a["pixels"]})
    assert got.shape[0] == want.shape[0] == a["input_ids"].shape[0]
    assert torch.equal(got[:, :a["input_ids"].shape[1]], a["input_ids"])
    g = torch.Generator().manual_seed(7)
    txt = torch.cat([torch.ones(3, 1, dtype=torch.long), torch.randint(3, 97, (3, 7), generator=g)], 1)
    got, want, ex_txt = run(txt, {}, {})
    assert int(ex_img.sum()) + int(ex_txt.sum()) >= (len(ex_img) + len(ex_txt) + 1) // 2       # most rows are the fp32 oracle's hypotheses outright
    # one beam is the greedy loop
    g1 = model.generate(txt.cuda(), modal_inputs={}, num_beams=1, max_new_tokens=n_new).cpu()
    b1 = beam.beam_search(lambda rows: om.prefill(rows, {}, last_only=True)[0][:, -1], txt, 1, n_new, eos, pad)
    n = min(g1.shape[1], b1.shape[1])
    assert torch.equal(g1[:, :n], b1[:, :n])


def test_ragged_and_text_only_batch_equals_per_sample_oracle(g4_model):
    """Edge cases of the splice / decode path: a batch whose samples have different spliced lengths (one of them text only, no image
    block).  Positions are per sample on the HIP path, so every sample must reproduce the oracle run on that sample alone
    (the reference's batched position ids ignore padding, SURVEY App. B, so batch-1 is the meaningful reference)."""
    from oracle import pipeline
    model, a, meta, sd = g4_model
    om = pipeline.OracleModel.from_state_dict(sd, meta)
    V = -200
    g = torch.Generator().manual_seed(77)
    r = lambda n: torch.randint(3, 97, (n,), generator=g).tolist()
    s0 = [1] + r(4) + [V, 13] + r(3)                 # image, length 10 -> spliced 9 + block
    s1 = [1] + r(9)                                  # text only, same text length, no block
    ids = torch.tensor([s0, s1], dtype=torch.long)
    px = a["pixels"][:1]
    n_new = 5
    res, lg = model.generate(ids.cuda(), modal_inputs={"vision": px.cuda()}, max_new_tokens=n_new, ignore_eos=True, return_step_logits=True)
    got = res[:, ids.shape[1]:].cpu()
    for b, (row, mi) in enumerate(((s0, {"vision": px}), (s1, {}))):
        ids_o, lg_o = om.generate(torch.tensor([row]), mi, max_new_tokens=n_new, ignore_eos=True, return_logits=True)
        within(f"9 ragged row {b}", lg[b:b + 1], lg_o, 1.1e-2)      # measured 5.4e-3
        assert torch.equal(got[b], ids_o[0]), (b, got[b], ids_o[0], _margins(lg_o))           # all 5 ids, no margin gate
    # text-only batch: no modal inputs at all (modal_inputs={} -> no routing, multimodal_llama.py:703-704)
    res2, lg2 = model.generate(torch.tensor([s1]).cuda(), modal_inputs={}, max_new_tokens=3, ignore_eos=True, return_step_logits=True)
    _, lg_o2 = om.generate(torch.tensor([s1]), {}, max_new_tokens=3, ignore_eos=True, return_logits=True)
    within("10 text only", lg2, lg_o2, 1.1e-2)


def test_limits_raise_like_the_reference(g4_model):
    model, a, meta, sd = g4_model
    ids = a["input_ids"].cuda()
    with pytest.raises(ValueError):                   # sentinel without an input for that modality
        model.generate(ids, modal_inputs={}, max_new_tokens=2)
    with pytest.raises(ValueError):                   # longer than the rotary table (max_position_embeddings)
        model.generate(ids, modal_inputs={"vision": a["pixels"].cuda()}, max_new_tokens=meta["max_position_embeddings"])
    out = model.forward(input_ids=ids, modal_inputs={"vision": a["pixels"].cuda()}, use_cache=True)
    with pytest.raises(NotImplementedError):          # per-layer outputs belong to the prefill call, a cached step returns logits only
        model.forward(input_ids=ids[:, -1:], past_key_values=out.past_key_values, output_hidden_states=True)
    # an id outside embed_tokens: nn.Embedding's IndexError, not a read past the table (round 6: a test tokenizer with ids >= vocab_size
    # had passed for rounds and faulted only when the neighbouring pages happened to be unmapped)
    bad = ids.clone()
    bad[0, 0] = meta["vocab_size"] + 7
    with pytest.raises(IndexError):
        model.generate(bad, modal_inputs={"vision": a["pixels"].cuda()}, max_new_tokens=2)
    with pytest.raises(IndexError):
        model.forward(input_ids=torch.full_like(ids[:, -1:], meta["vocab_size"]), past_key_values=out.past_key_values)


def test_load_pretrained_model_from_checkpoint_directories(tmp_path):
    """A15: the loader reads the same files as builder.py:138-185 — base shards (sharded .bin with index), adapter_model.bin,
    non_lora_trainables.bin, config.json, the CLIP directory — and the loaded model reproduces the in-memory build bit for bit."""
    import json
    import os
    from modelcompose_amd.model.builder import build_from_state_dict, load_pretrained_model
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a, meta, sd = load_golden("g4_e2e_vision")
    clip_dir = tmp_path / "clip-tiny"
    clip_dir.mkdir()
    pre = "model.modal_encoders.vision.vision_tower."
    torch.save({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, clip_dir / "pytorch_model.bin")
    json.dump(dict(meta["clip"], model_type="clip_vision_model"), open(clip_dir / "config.json", "w"))
    json.dump({"crop_size": 28, "size": 28, "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711]},
              open(clip_dir / "preprocessor_config.json", "w"))
    base, ckpt = tmp_path / "vicuna-tiny", tmp_path / "multimodal-tiny-lora"
    base.mkdir(); ckpt.mkdir()
    is_adapter = lambda k: ".lora_" in k or k.startswith("prefix_tokens") or k.startswith("suffix_tokens")
    is_nonlora = lambda k: k.startswith("model.modal_projectors.")
    base_sd = {k: v for k, v in sd.items() if not (is_adapter(k) or is_nonlora(k) or k.startswith(pre))}
    keys = sorted(base_sd)
    half = len(keys) // 2
    shards = {"pytorch_model-00001-of-00002.bin": keys[:half], "pytorch_model-00002-of-00002.bin": keys[half:]}
    for fn, ks in shards.items():
        torch.save({k: base_sd[k] for k in ks}, base / fn)
    json.dump({"weight_map": {k: fn for fn, ks in shards.items() for k in ks}}, open(base / "pytorch_model.bin.index.json", "w"))
    torch.save({"base_model.model." + k: v for k, v in sd.items() if is_adapter(k)}, ckpt / "adapter_model.bin")
    torch.save({k: v for k, v in sd.items() if is_nonlora(k)}, ckpt / "non_lora_trainables.bin")
    cfg = {k: v for k, v in meta.items() if k not in ("clip", "modal_names")}
    cfg["mm_vision_encoder"] = str(clip_dir)
    json.dump(cfg, open(ckpt / "config.json", "w"))
    tok, model, procs, ctx = load_pretrained_model(str(ckpt), str(base), "multimodal-tiny-lora")
    assert ctx == 2048 and set(procs) == {"vision"} and procs["vision"] is not None
    ref = build_from_state_dict(meta, sd)
    ids, px = a["input_ids"].cuda(), a["pixels"].cuda()
    out1 = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=6, ignore_eos=True)
    out2 = ref.generate(ids, modal_inputs={"vision": px}, max_new_tokens=6, ignore_eos=True)
    assert torch.equal(out1, out2)
    with pytest.raises(ValueError):
        load_pretrained_model(str(ckpt), str(base), "llava-v1.5")            # only the 'multimodal' branch (builder.py:138)


def test_eval_model_driver_writes_the_answers_file_and_shards_by_chunk(tmp_path, g4_model):
    """The eval loop of model_multimodal_qa_loader.py on the HIP path: question file -> prompts (v1 template, sentinel tokenisation) ->
    collator (HIP image processor, pad-to-square) -> greedy generate -> answers.jsonl; 2 chunks concatenated == 1 chunk (rank k
    evaluates chunk k, the reference's data-parallel scheme)."""
    import json
    from types import SimpleNamespace
    from PIL import Image
    from modelcompose_amd.eval.model_multimodal_qa_loader import eval_model
    from modelcompose_amd.model.image_processor import HipCLIPImageProcessor
    from oracle.toy_tokenizer import ToyTokenizer
    model, a, meta, sd = g4_model
    rng = np.random.default_rng(0)
    files = []
    for i, (w, h) in enumerate(((40, 30), (28, 28), (25, 50))):
        f = tmp_path / f"img{i}.png"
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(f)
        files.append(str(f))
    qs = [{"id": f"q{i}", "conversations": [{"from": "human", "value": ("<image>\n" if i != 2 else "") + f"what is item {i} ?"}, {"from": "gpt", "value": ""}],
           **({"modal_inputs": {"vision": [files[min(i, 2)]]}} if i != 2 else {})} for i in range(4)]
    qfile = tmp_path / "questions.json"
    json.dump(qs, open(qfile, "w"))
    procs = {"vision": HipCLIPImageProcessor(size=28, crop_size=28)}

    def run(num_chunks, chunk_idx, out, pipeline=False):
        tok = ToyTokenizer(True, model_max_length=256)
        args = SimpleNamespace(model_path="/ckpts/multimodal-tiny", model_base=None, question_file=str(qfile), answers_file=str(out), conv_mode="v1",
                               num_chunks=num_chunks, chunk_idx=chunk_idx, temperature=0.0, top_p=None, num_beams=1, batch_size=1, max_new_tokens=5,
                               pipeline=pipeline)
        n = eval_model(args, loaded=(tok, model, procs, 2048))
        return n, [json.loads(l) for l in open(out)]

    n, full = run(1, 0, tmp_path / "all.jsonl")
    assert n == 4 and [r["question_id"] for r in full] == ["q0", "q1", "q2", "q3"]
    assert all(r["model_id"] == "multimodal-tiny" and r["prompt"] == q["conversations"][0]["value"] for r, q in zip(full, qs))
    np_, piped = run(1, 0, tmp_path / "piped.jsonl", pipeline=True)
    assert np_ == 4 and [(r["question_id"], r["text"]) for r in piped] == [(r["question_id"], r["text"]) for r in full]
    n0, part0 = run(2, 0, tmp_path / "c0.jsonl")
    n1, part1 = run(2, 1, tmp_path / "c1.jsonl")
    assert n0 == 2 and n1 == 2
    # a fresh toy tokenizer per run assigns ids in order of first appearance: chunk 1 alone sees different ids, so compare chunk 0 exactly
    # and chunk 1 structurally
    assert [r["text"] for r in part0] == [r["text"] for r in full[:2]]
    assert [r["question_id"] for r in part0 + part1] == [r["question_id"] for r in full]


def test_batched_eval_answers_do_not_depend_on_the_number_of_chunks(tmp_path, g4_model):
    """VERDICT r5 #1a / SURVEY §8(e): `--batch-size 4` answers of a 1-rank run and of a 3-rank run (chunk = ceil(n / N), rank k evaluates
    chunk k: model_multimodal_qa_loader.py:25-46; the answer files are concatenated: MCUB-4.sh:60-70) are identical row for row although
    every question sits in a different batch (4 + 3 rows against 3 + 3 + 1), with other neighbours, another padded length and another KV
    cache stride - prefill (tile GEMM family), decode steps (strip GEMM family, per-sequence attention chunks) and the lm_head are all
    batch-invariant.  Generated ids are compared bitwise through generate() as well."""
    import json
    from types import SimpleNamespace
    from PIL import Image
    from modelcompose_amd.eval.model_multimodal_qa_loader import eval_model
    from modelcompose_amd.model.image_processor import HipCLIPImageProcessor
    from oracle.toy_tokenizer import ToyTokenizer
    model, a, meta, sd = g4_model
    rng = np.random.default_rng(1)
    files = []
    for i, (w, h) in enumerate(((40, 30), (28, 28), (25, 50), (33, 21))):
        f = tmp_path / f"img{i}.png"
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(f)
        files.append(str(f))
    texts = ["what is item 0 ?", "describe this picture in a few words please", "no picture here , just text", "what colour ?", "how many things are there in it ?",
             "a", "is it item 6 or item 7 or neither ?"]
    qs = [{"id": f"q{i}", "conversations": [{"from": "human", "value": ("<image>\n" if i != 2 else "") + t}, {"from": "gpt", "value": ""}],
           **({"modal_inputs": {"vision": [files[i % 4]]}} if i != 2 else {})} for i, t in enumerate(texts)]
    qfile = tmp_path / "questions7.json"
    json.dump(qs, open(qfile, "w"))
    procs = {"vision": HipCLIPImageProcessor(size=28, crop_size=28)}
    tok = ToyTokenizer(True, model_max_length=256)                      # ONE tokenizer: the toy assigns ids in order of first appearance

    def run(num_chunks, chunk_idx, out, bs):
        args = SimpleNamespace(model_path="/ckpts/multimodal-tiny", model_base=None, question_file=str(qfile), answers_file=str(out), conv_mode="v1",
                               num_chunks=num_chunks, chunk_idx=chunk_idx, temperature=0.0, top_p=None, num_beams=1, batch_size=bs, max_new_tokens=6,
                               pipeline=False)
        n = eval_model(args, loaded=(tok, model, procs, 2048))
        return [json.loads(l) for l in open(out)]

    one = run(1, 0, tmp_path / "n1.jsonl", 4)
    assert [r["question_id"] for r in one] == [f"q{i}" for i in range(7)]
    three = sum((run(3, k, tmp_path / f"n3_{k}.jsonl", 4) for k in range(3)), [])
    assert [(r["question_id"], r["text"]) for r in three] == [(r["question_id"], r["text"]) for r in one]
    alone = run(1, 0, tmp_path / "b1.jsonl", 1)                        # the reference's own geometry: one question per generate() call
    assert [(r["question_id"], r["text"]) for r in alone] == [(r["question_id"], r["text"]) for r in one]


def test_generate_pipelined_equals_sequential_generate(g4_model):
    """Two alternating generation pipelines on two streams produce exactly the tokens of sequential generate() calls, batch after
    batch (different prompts / images per batch, greedy and sampled, with and without EOS handling)."""
    model, a, meta, sd = g4_model
    g = torch.Generator().manual_seed(3)
    batches = []
    for i in range(5):
        ids = a["input_ids"].clone()
        ids[:, 1:4] = torch.randint(3, 100, (ids.shape[0], 3), generator=g)
        px = (a["pixels"] + 0.1 * i).cuda()
        batches.append((ids.cuda(), {"vision": px}))
    for kw in (dict(max_new_tokens=6, ignore_eos=True), dict(max_new_tokens=6), dict(max_new_tokens=5, ignore_eos=True, do_sample=True, temperature=1.3, seed=7)):
        seq = [model.generate(ids, modal_inputs=mi, **kw) for ids, mi in batches]
        pip = list(model.generate_pipelined(iter(batches), **kw))
        assert len(pip) == len(seq)
        for x, y in zip(seq, pip):
            assert torch.equal(x, y)
    # and the plain path still works afterwards (slot 0 buffers reused)
    again = model.generate(batches[0][0], modal_inputs=batches[0][1], max_new_tokens=5, ignore_eos=True, do_sample=True, temperature=1.3, seed=7)
    assert torch.equal(again, seq[0])


def test_decode_graph_is_really_replayed_on_the_default_stream(g4_model):
    """hipStreamBeginCapture is refused on the legacy null stream (torch's default current stream).  The runtime captures
    and replays on its own stream then; the query must say a graph ran, and the tokens must equal the one-launch-per-kernel path's."""
    from modelcompose_amd import _lib
    model, a, meta, sd = g4_model
    ids, px = a["input_ids"].cuda(), a["pixels"].cuda()
    assert torch.cuda.current_stream().cuda_stream == 0
    fails0 = model.runtime_option("graph_failures")
    out_g = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=8, ignore_eos=True)
    assert model.runtime_option("graph_active") == 1 and model.runtime_option("graph_failures") == fails0
    # results are visible to work queued on the default stream afterwards without a host sync (event hand-back)
    first = out_g[:, ids.shape[1]:].clone()
    _lib.check(_lib.lib().mc_llm_set_option(model._handle, b"use_graph", 0), "set_option")
    try:
        out_e = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=8, ignore_eos=True)
        assert model.runtime_option("graph_active") == 0
    finally:
        _lib.check(_lib.lib().mc_llm_set_option(model._handle, b"use_graph", 1), "set_option")
    assert torch.equal(first, out_e[:, ids.shape[1]:])
    # a caller-owned stream is captured on directly
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out_s = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=8, ignore_eos=True)
        assert model.runtime_option("graph_active") == 1
    s.synchronize()
    assert torch.equal(out_s, out_g)


def test_right_padded_batch_equals_per_sample(g4_model):
    """A batch right-padded by the collator with its attention mask: every row generates exactly what it generates alone (the device
    path takes one length per sequence)."""
    model, a, meta, sd = g4_model
    V = -200
    g = torch.Generator().manual_seed(5)
    r = lambda n: torch.randint(3, 97, (n,), generator=g).tolist()
    rows = [[1] + r(3) + [V, 13] + r(6), [1] + r(2) + [V, 13] + r(3), [1] + r(7)]
    L = max(len(x) for x in rows)
    ids = torch.tensor([x + [0] * (L - len(x)) for x in rows])
    am = torch.tensor([[1] * len(x) + [0] * (L - len(x)) for x in rows], dtype=torch.bool)
    px = torch.cat([a["pixels"][:1], a["pixels"][:1] * 0.5])
    n_new = 5
    res, lg = model.generate(ids.cuda(), modal_inputs={"vision": px.cuda()}, attention_mask=am.cuda(), max_new_tokens=n_new, ignore_eos=True,
                             return_step_logits=True)
    for b, row in enumerate(rows):
        mi = {"vision": px[b:b + 1].cuda()} if V in row else {}
        r1, l1 = model.generate(torch.tensor([row]).cuda(), modal_inputs=mi, max_new_tokens=n_new, ignore_eos=True, return_step_logits=True)
        assert torch.equal(l1[0, 0], lg[b, 0]), b                        # prefill logits: bitwise (batch-invariant kernels)
        assert torch.equal(r1[0, len(row):], res[b, L:]), b


def _oracle_of(meta, sd):
    from oracle import pipeline
    sdf = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    return pipeline.OracleModel.from_state_dict(sdf, meta)


def test_left_padded_and_holey_masks_follow_the_reference(g4_model):
    """VERDICT r2 missing #4: attention masks whose zeros are not a suffix (multimodal_llama.py:526-531, :543-545: additive padding mask,
    position ids ignore padding), against the oracle's restatement of the reference loop, teacher-forced so that every step compares:
      (a) text-only left-padded batch, generate(modal_inputs=None): the mask hides the pad KEYS in the prefill and - extended by ones - in
          every decode step, exactly the reference (multimodal_arch.py:290-293 returns the caller's mask untouched when modal_inputs is None);
      (b) the same with modal_inputs={} and (c) a left-padded batch WITH an image and a hole in the mask: the reference left-extends the
          mask by the inserted length (:445-449), so the zeros land on the first block tokens instead of the pads - reproduced through the
          splice plan.  On decode steps the reference would replace the mask by all ones here (:292); the HIP path keeps it (DESIGN §7:
          the unmasked pads hold the outputs of rows that saw no key - garbage by construction), so (b) / (c) are compared with the oracle
          run with keep_mask=True, and (b) must equal (a)."""
    model, a, meta, sd = g4_model
    om = _oracle_of(meta, sd)
    g = torch.Generator().manual_seed(9)
    r = lambda n: torch.randint(3, 97, (n,), generator=g).tolist()
    n_new = 5

    def run(ids_, am_, mi_hip, mi_or, keep=None):
        with torch.no_grad():
            ids_o, lg_o = om.generate(ids_, mi_or, max_new_tokens=n_new, ignore_eos=True, return_logits=True, attention_mask=am_, keep_mask=keep)
        res, lg = model.generate(ids_.cuda(), modal_inputs=mi_hip, attention_mask=am_.cuda(), max_new_tokens=n_new, ignore_eos=True,
                                 return_step_logits=True, forced_ids=ids_o[:, :n_new - 1])
        lg = lg.float().cpu()
        sc = lg_o.abs().max()
        err = ((lg - lg_o).abs().max() / sc).item()
        got = res[:, ids_.shape[1]:].cpu()
        top2 = lg_o.topk(2, -1).values
        clear = ((top2[..., 0] - top2[..., 1]) / sc) > 2e-2                 # steps whose oracle margin is well above the logit error
        assert err < 1.5e-2, ((lg - lg_o).abs().amax(-1) / sc)
        assert torch.equal(got[clear], ids_o[clear]), (got, ids_o, clear)
        return lg, lg_o, err, sc
    rows = [[1] + r(9), [1] + r(4), [1] + r(6)]
    L = max(len(x) for x in rows)
    ids = torch.tensor([[0] * (L - len(x)) + x for x in rows])
    am = torch.tensor([[0] * (L - len(x)) + [1] * len(x) for x in rows], dtype=torch.bool)
    lg_a, lg_oa, err, scale = run(ids, am, None, None)
    print(f"[err] left padding (a): {err:.3e}")
    # the pads are really hidden: the first-token logits of every row equal those of the row alone, un-padded (RoPE is relative: shifting
    # a whole sequence changes nothing but rounding)
    for b, row in enumerate(rows):
        r1, l1 = model.generate(torch.tensor([row]).cuda(), modal_inputs=None, max_new_tokens=1, ignore_eos=True, return_step_logits=True)
        assert ((l1[0, 0].float().cpu() - lg_a[b, 0]).abs().max() / scale).item() < 1.5e-2, b
    # and hiding them matters: without the mask rows 1 and 2 compute something else
    _, lg_nomask = model.generate(ids.cuda(), modal_inputs=None, max_new_tokens=1, ignore_eos=True, return_step_logits=True)
    assert ((lg_nomask.float().cpu()[1:, 0] - lg_a[1:, 0]).abs().max() / scale).item() > 5e-2
    lg_b, _, err, _ = run(ids, am, {}, {}, keep=True)
    print(f"[err] left padding (b): {err:.3e}")
    assert torch.equal(lg_b, lg_a)                                              # same function as (a) on the HIP path
    # (c) image + left padding, and a hole
    V = -200
    rows = [[1] + r(3) + [V, 13] + r(6), [1] + r(2) + [V, 13] + r(3)]
    L = max(len(x) for x in rows)
    ids = torch.tensor([[0] * (L - len(x)) + x for x in rows])
    am = torch.tensor([[0] * (L - len(x)) + [1] * len(x) for x in rows], dtype=torch.bool)
    am[0, 5] = False                                                            # a hole in the longest row
    px = torch.cat([a["pixels"][:1], a["pixels"][:1] * 0.5])
    _, _, err, _ = run(ids, am, {"vision": px.cuda()}, {"vision": px.float()}, keep=True)
    print(f"[err] left padding + image + hole (c): {err:.3e}")
    # forward() with such a mask: logits of the last positions against the oracle's prefill
    out = model(input_ids=ids.cuda(), attention_mask=am.cuda(), modal_inputs={"vision": px.cuda()})
    with torch.no_grad():
        lo, _, _ = om.prefill(ids, {"vision": px.float()}, attention_mask=am)
    assert out.logits.shape == lo.shape
    e2 = ((out.logits.float().cpu() - lo).abs().amax(-1) / lo.abs().max())
    assert float(e2[:, -8:].max()) < 1.5e-2


def test_left_padded_batch_of_mixed_spliced_lengths_equals_each_row_alone(g4_model):
    """a left-padded batch whose rows carry different numbers of modal tokens (an image row + a text-only row).  The
    shorter row's spliced length is below Lmax, the splice right-pads its mask with zeros over [lens[b], Lmax), and decode appends that
    row's generated tokens exactly there: those slots must be open as keys.  Every row must generate what it generates alone (same
    tokens teacher-forced; logits to rounding - RoPE is relative, the shift only changes rounding)."""
    model, a, meta, sd = g4_model
    g = torch.Generator().manual_seed(21)
    r = lambda n: torch.randint(3, 97, (n,), generator=g).tolist()
    V = -200
    rows = [[1] + r(3) + [V, 13] + r(6), [1] + r(5)]                      # image row, text-only row
    L = max(len(x) for x in rows)
    ids = torch.tensor([[0] * (L - len(x)) + x for x in rows])
    am = torch.tensor([[0] * (L - len(x)) + [1] * len(x) for x in rows], dtype=torch.bool)
    px = a["pixels"][:1]
    n_new = 6
    alone = []
    for row in rows:
        mi = {"vision": px.cuda()} if V in row else {}
        r1, l1 = model.generate(torch.tensor([row]).cuda(), modal_inputs=mi, max_new_tokens=n_new, ignore_eos=True, return_step_logits=True)
        alone.append((r1[0, len(row):].cpu(), l1[0].float().cpu()))
    forced = torch.stack([x[0][:n_new - 1] for x in alone])
    res, lg = model.generate(ids.cuda(), modal_inputs={"vision": px.cuda()}, attention_mask=am.cuda(), max_new_tokens=n_new, ignore_eos=True,
                             return_step_logits=True, forced_ids=forced)
    lg = lg.float().cpu()
    for b in range(len(rows)):
        sc = alone[b][1].abs().max()
        err = ((lg[b] - alone[b][1]).abs().amax(-1) / sc)
        print(f"[err] mixed-length left padding, row {b}: {err.tolist()}")
        assert float(err.max()) < 1.5e-2, (b, err)                        # before the fix: the text row lost sight of its own new tokens
        top2 = alone[b][1].topk(2, -1).values
        clear = ((top2[:, 0] - top2[:, 1]) / sc) > 2e-2
        assert torch.equal(res[b, L:].cpu()[clear], alone[b][0][clear]), b


def test_last_layer_tail_equals_the_full_last_layer(g4_model):
    """generate()'s prefill runs the last decoder layer's attention + MLP for the last token of every sequence only (mc_llm option
    "tail_adapter"): the first-step logits agree with the all-rows path to fp32 summation order of one attention row and three small GEMMs
    (<= 4e-3 of the logit scale - both are within 1e-2 of the reference), the greedy ids are the same, the KV cache of the last layer is
    BIT-identical (its q|k|v projection still covers every row), and forward() - which needs every row - is untouched; ragged batch too."""
    model, a, meta, sd = g4_model
    ids = a["input_ids"].cuda()
    px = a["pixels"].cuda()
    n_new = a["gen_ids"].shape[1]
    outs = {}
    for flag in (False, True):
        model.last_layer_tail = flag
        try:
            res, lg = model.generate(ids, modal_inputs={"vision": px}, max_new_tokens=n_new, ignore_eos=True, return_step_logits=True)
            torch.cuda.synchronize()
            kc = [t.clone() for t in model._cache[next(k for k in model._cache if isinstance(k, tuple) and k and k[0] == "kv")]] \
                if any(isinstance(k, tuple) and k and k[0] == "kv" for k in model._cache) else None
            fw = model.forward(input_ids=ids, modal_inputs={"vision": px}).logits.clone()
        finally:
            del model.last_layer_tail
        outs[flag] = (res.cpu(), lg.float().cpu(), kc, fw.float().cpu())
    assert torch.equal(outs[True][0], outs[False][0])
    assert torch.equal(outs[True][0][:, ids.shape[1]:], a["gen_ids"])
    scale = a["step_logits"].abs().max().item()
    d = (outs[True][1] - outs[False][1]).abs().max().item() / scale
    print(f"[err] tail vs full last layer, step logits: {d:.3e}")
    assert d <= 4e-3, d
    within("tail step logits vs reference", outs[True][1], a["step_logits"], 1e-2)
    assert torch.equal(outs[True][3], outs[False][3])                     # forward(): every row, never the tail
    if outs[True][2] is not None:
        for x, y in zip(outs[True][2], outs[False][2]):
            assert torch.equal(x, y)

