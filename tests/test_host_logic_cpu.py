"""Host-side integer logic of the product (no GPU): splice planner bit-exact against the reference's golden masks /
labels / attention masks, adapter plan, routing layout, partition rule, merge CLI."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from modelcompose_amd.model import config as mcfg
from modelcompose_amd.model.splice import plan_splice, routed_layout


def _plan_case(a, tag, keys, block_len, n_items):
    ids = a[f"{tag}::input_ids"].numpy()
    am = a[f"{tag}::attention_mask_in"].numpy()
    lab = a[f"{tag}::labels_in"].numpy() if f"{tag}::labels_in" in a else None
    return plan_splice(ids, am, lab, keys, block_len, n_items)


def test_plan_splice_matches_reference_masks_labels():
    a, _, _ = load_golden("g3_splice")
    cases = [("eq", ["vision", "audio"], {"vision": 4 + 3, "audio": 3 + 3}, {"vision": 2, "audio": 2}),
             ("ragged", ["vision", "video"], {"vision": 4 + 3, "video": 6 + 3}, {"vision": 2, "video": 1}),
             ("edge", ["vision", "video"], {"vision": 4, "video": 6}, {"vision": 2, "video": 2})]
    for tag, keys, bl, ni in cases:
        p = _plan_case(a, tag, keys, bl, ni)
        assert np.array_equal(p.attention_mask, a[f"{tag}::attention_mask"].numpy()), tag
        if p.labels is not None:
            assert np.array_equal(p.labels, a[f"{tag}::labels"].numpy()), tag
        exp = {k.split("::")[-1]: v.numpy() for k, v in a.items() if k.startswith(f"{tag}::mask::")}
        assert set(p.modal_masks) == set(exp)
        for k in exp:
            assert np.array_equal(p.modal_masks[k], exp[k]), (tag, k)
        # text rows index the embedding table
        emb = a[f"{tag}::embeds"]
        table = a["embed_tokens"]
        B, Lmax = p.tok_id.shape
        for b in range(B):
            for t in range(int(p.lens[b])):
                if p.src_modal[b, t] < 0:
                    assert torch.equal(emb[b, t], table[p.tok_id[b, t]])


def test_plan_splice_errors():
    ids = np.array([[1, -200, 5]])
    with pytest.raises(ValueError):
        plan_splice(ids, None, None, [], {}, {})                       # sentinel without a modal_inputs entry
    with pytest.raises(ValueError):
        plan_splice(np.array([[1, -200, -200]]), None, None, ["vision"], {"vision": 3}, {"vision": 1})   # not enough items


def test_routed_layout_is_a_permutation_grouped_by_adapter():
    ids = np.array([[1, 7, -200, 13, 8, -203, 9], [1, -203, 13, 7, 8, -200, 9]])
    p = plan_splice(ids, None, None, ["vision", "audio"], {"vision": 5, "audio": 2}, {"vision": 2, "audio": 2})
    lay = routed_layout(p, {"vision": 2, "audio": 1}, routed=True)
    assert lay.M == int(p.lens.sum())
    assert sorted(lay.out_map[lay.out_map >= 0].tolist()) == list(range(lay.M))
    assert lay.group_adapter.tolist() == [0, 1, 2]
    for g, (s, e) in enumerate(zip(lay.group_start[:-1], lay.group_start[1:])):
        for r in range(s, e):
            b, t = lay.order_b[r], lay.order_t[r]
            exp = {-1: 0, p.modal_order.index("vision"): 2, p.modal_order.index("audio"): 1}[int(p.src_modal[b, t])]
            assert exp == lay.group_adapter[g]
    for b in range(2):
        r = lay.last_rows[b]
        assert lay.order_b[r] == b and lay.order_t[r] == p.lens[b] - 1
    un = routed_layout(p, {"vision": 2, "audio": 1}, routed=False)
    assert un.group_adapter.tolist() == [0] and un.M == lay.M


def test_adapter_plan_matches_reference_fixture():
    _, meta, _ = load_golden("g1_lora_linear")
    cfg = mcfg.MultimodalConfig(lora_r=meta["lora_r"], lora_alpha=meta["lora_alpha"], mm_audio_encoder="a", mm_vision_encoder="v",
                                mm_video_encoder="d", reset_scaling_weights=meta["reset_scaling_weights"])
    names, scaling, dn, merge = mcfg.adapter_plan(cfg)
    assert names == meta["adapters"] and dn == meta["default_adapter_names"] and merge == meta["merge_default_weights"]
    for k, v in meta["scaling"].items():
        assert abs(scaling[k] - v) < 1e-12
    terms = mcfg.composition_terms(cfg, "default", lambda k: True)
    assert [t[0] for t in terms] == dn
    assert mcfg.composition_terms(cfg, "point", lambda k: True) == []
    assert mcfg.composition_terms(cfg, "vision", lambda k: k != "vision") == []
    with pytest.raises(ValueError):
        mcfg.extract_params("default-vision")                             # malformed strategy string
    with pytest.raises(ValueError):
        mcfg.MultimodalConfig(rope_scaling={"type": "linear", "factor": 2})


def test_partition_rule_matches_reference():
    from modelcompose_amd.dist import get_chunk, split_list
    lst = list(range(10))
    assert split_list(lst, 4) == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]
    assert get_chunk(lst, 8, 1) == [2, 3]
    assert sum(split_list(lst, 3), []) == lst


def test_merge_cli_matches_reference_golden(tmp_path):
    from modelcompose_amd import compose
    a, meta, _ = load_golden("g6_merge")
    paths = []
    for modal in meta["order"]:
        d = tmp_path / f"ckpt-{modal}"
        d.mkdir()
        torch.save({k.split("::", 2)[2]: v for k, v in a.items() if k.startswith(f"in::{modal}::")}, d / "adapter_model.bin")
        json.dump(meta["in_configs"][modal], open(d / "config.json", "w"))
        paths.append(str(d))
    out = tmp_path / "merged"
    compose.main(paths + ["-o", str(out), "--strategy", meta["strategy"]])
    got = torch.load(out / "adapter_model.bin")
    exp = {k[5:]: v for k, v in a.items() if k.startswith("out::")}
    assert sorted(got) == sorted(exp) and all(torch.equal(got[k], exp[k]) for k in exp)
    assert json.load(open(out / "config.json")) == meta["out_config"]
    assert open(out / "merge_info.txt").read().replace(str(tmp_path), "<TMP>") == meta["merge_info"]
    with pytest.raises(AssertionError):           # convert-* is for checkpoints trained with lora_strategy 'same' (:47)
        compose.merge_checkpoints(paths, str(out), "convert-ties-sum")
    with pytest.raises(NotImplementedError):
        compose.merge_checkpoints(paths, str(out), "slerp")
    if not torch.cuda.is_available():             # TIES runs on the HIP device and fails loudly without one (no CPU fallback)
        with pytest.raises(RuntimeError):
            compose.merge_checkpoints(paths, str(out), "ties-sum")


def test_length_grouped_samplers_match_reference():
    """Index-exact against the reference's llava_trainer samplers (tests/golden/g12_host.npz) for the same generator seeds."""
    from conftest import load_golden
    from modelcompose_amd.train import sampler
    _, meta, _ = load_golden("g12_host")
    for c in meta["chunks"]:
        assert sampler.split_to_even_chunks(c["indices"], c["lengths"], c["num_chunks"]) == c["chunks"]
    for c in meta["plain"]:
        g = torch.Generator().manual_seed(c["seed"])
        got = sampler.get_length_grouped_indices(c["lengths"], c["batch_size"], c["world_size"], generator=g)
        assert got == c["indices"] and sorted(got) == list(range(len(c["lengths"])))
    for c in meta["modality"]:
        torch.manual_seed(c["global_seed"])
        g = torch.Generator().manual_seed(c["seed"])
        s = sampler.LengthGroupedSampler(c["batch_size"], c["world_size"], lengths=c["lengths"], generator=g, group_by_modality=True)
        assert list(s) == c["indices"] and len(s) == len(c["lengths"])
    with pytest.raises(ValueError):
        sampler.LengthGroupedSampler(2, 2)
    with pytest.raises(AssertionError):
        sampler.get_modality_length_grouped_indices([3, 0, -2], 1, 1)


def test_convert_llava_checkpoint_matches_reference(tmp_path):
    """Key mapping and file split of convert_checkpoint.py, byte-exact tensors, only the whitelisted side files copied."""
    import json
    from conftest import load_golden
    from modelcompose_amd import compose
    a, meta, _ = load_golden("g12_host")
    src = tmp_path / "llava"
    src.mkdir()
    for fn in ("pytorch_model-00001-of-00002.bin", "pytorch_model-00002-of-00002.bin"):
        torch.save({k.split("::", 2)[2]: v.to(torch.float16) for k, v in a.items() if k.startswith(f"in::{fn}::")}, src / fn)
    json.dump({"model_type": "llava", "hidden_size": 8}, open(src / "config.json", "w"))
    (src / "tokenizer_config.json").write_text("{}")
    (src / "unrelated.txt").write_text("x")
    out = tmp_path / "out"
    compose.convert_llava_checkpoint(str(src), str(out))
    assert sorted(p.name for p in out.iterdir()) == meta["convert_files"]
    for fn in ("adapter_model.bin", "non_lora_trainables.bin"):
        got = torch.load(out / fn)
        exp = {k.split("::", 2)[2]: v for k, v in a.items() if k.startswith(f"out::{fn}::")}
        assert sorted(got) == sorted(exp)
        for k in exp:
            assert got[k].dtype == torch.float16 and torch.equal(got[k].float(), exp[k]), k
    assert compose.llava_key_to_multimodal_key("model.layers.0.self_attn.q_proj.weight") is None


def test_prompt_helpers_match_reference_mm_utils():
    """tokenizer_image_token / tokenizer_modal_token / split_string_by_list / KeywordsStoppingCriteria / expand2square /
    get_model_name_from_path against the outputs of the reference's own mm_utils.py (tests/golden/g13_prompt.npz)."""
    from PIL import Image
    from conftest import load_golden
    from modelcompose_amd import mm_utils
    from modelcompose_amd.constants import MODAL_TOKEN_MAPPING
    from oracle.toy_tokenizer import ToyTokenizer
    a, meta, _ = load_golden("g13_prompt")
    for c in meta["image_token"]:
        assert mm_utils.tokenizer_image_token(c["prompt"], ToyTokenizer(c["add_bos"])) == c["ids"], c
    for c in meta["modal_token"]:
        assert mm_utils.tokenizer_modal_token(c["prompt"], ToyTokenizer(c["add_bos"])) == c["ids"], c
    t = mm_utils.tokenizer_modal_token(meta["prompts"][3], ToyTokenizer(True), return_tensors="pt")
    assert t.dtype == torch.int64 and sorted(int(x) for x in t[t < 0]) == [-205, -204, -203, -200]
    with pytest.raises(ValueError):
        mm_utils.tokenizer_modal_token("x", ToyTokenizer(True), return_tensors="np")
    for c in meta["split"]:
        assert [list(x) for x in mm_utils.split_string_by_list(c["prompt"], list(MODAL_TOKEN_MAPPING.keys()))] == c["out"]
    tok = ToyTokenizer(True)
    prompt_ids = torch.tensor([tok("USER: hello there ASSISTANT:").input_ids])
    for c in meta["stop"]:
        crit = mm_utils.KeywordsStoppingCriteria(c["keywords"], tok, prompt_ids)
        new = tok(c["text"]).input_ids[1:]
        got = [bool(crit(torch.cat([prompt_ids, torch.tensor([new[:n]])], dim=1), None)) for n in range(1, len(new) + 1)]
        assert got == c["verdicts"], c
    for c in meta["names"]:
        assert mm_utils.get_model_name_from_path(c["path"]) == c["name"]
    for name in ("wide", "tall", "square"):
        out = mm_utils.expand2square(Image.fromarray(a[f"img::{name}"].numpy().astype(np.uint8)), (12, 200, 77))
        assert np.array_equal(np.asarray(out).astype(np.int32), a[f"sq::{name}"].numpy())


def test_preprocess_and_collator_match_reference_data_utils():
    """conversation -> (input_ids, labels) for the v1 / llama_2 / plain / v0 templates and the batch collator, element-exact against
    the reference's data/utils.py and multimodal_dataset.py outputs (tests/golden/g14_data.npz)."""
    import copy
    from PIL import Image
    from conftest import load_golden
    from modelcompose_amd import conversation as cl
    from modelcompose_amd import data as mdata
    from oracle.toy_tokenizer import FakeProc, ToyTokenizer
    a, meta, _ = load_golden("g14_data")
    convs = meta["convs"]
    saved = cl.default_conversation
    try:
        for c in meta["cases"]:
            cl.default_conversation = cl.conv_templates[c["template"]]
            tok = ToyTokenizer(True, model_max_length=c["max_length"])
            out = mdata.preprocess([copy.deepcopy(convs[c["conv"]])], tok, has_image=c["has_image"])
            assert torch.equal(out["input_ids"][0], a[f"ids::{c['n']}"]), c
            assert torch.equal(out["labels"][0], a[f"labels::{c['n']}"]), c
        cl.default_conversation = cl.conv_templates["v1"]
        tok = ToyTokenizer(True, model_max_length=meta["collate"]["max_length"])
        imgs = [a[f"col::img::{i}"].numpy().astype(np.uint8) for i in range(3)]
        vids = [a[f"col::vid::{i}"] for i in range(2)]
        pts = [a["col::pts::0"].numpy()]
        data = []
        for cname, mi in meta["collate"]["samples"]:
            item = {"conversations": copy.deepcopy(convs[cname]), "modal_inputs": {}}
            for k, v in mi.items():
                item["modal_inputs"][k] = ([Image.fromarray(imgs[i]) for i in v] if k == "vision" else [vids[i] for i in v] if k == "video"
                                           else [pts[i] for i in v] if k == "point" else list(v))
            data.append(item)
        ds = mdata.MultimodalDataset(data, tok)
        assert len(ds) == 4
        procs = {"vision": FakeProc("vision"), "audio": FakeProc("audio"), "point": FakeProc("point"), "video": None}
        batch = mdata.DataCollatorForSupervisedDataset(tok, procs, {"vision": {"image_aspect_ratio": "pad"}})([ds[i] for i in range(4)])
    finally:
        cl.default_conversation = saved
    assert torch.equal(batch["input_ids"], a["col::input_ids"]) and torch.equal(batch["labels"], a["col::labels"])
    assert torch.equal(batch["attention_mask"].to(torch.int32), a["col::attention_mask"])
    assert list(batch["modal_inputs"].keys()) == meta["collate"]["modal_keys"]
    # the stand-in vision processor has no pad_to_square; compare through the reference's expand2square semantics instead
    assert torch.equal(batch["modal_inputs"]["video"], a["col::video"]) and torch.equal(batch["modal_inputs"]["point"], a["col::point"])
    assert torch.equal(batch["modal_inputs"]["audio"]["audio_inputs"], a["col::audio_inputs"])
    assert torch.equal(batch["modal_inputs"]["audio"]["audio_padding_mask"].to(torch.int32), a["col::audio_padding_mask"])
    assert batch["modal_inputs"]["vision"].shape == a["col::vision"].shape
    # modality_lengths: language-only negative, +256 per vision sample, +257*8 per video clip
    ds2 = mdata.MultimodalDataset([{"conversations": convs["text_only"]}, {"conversations": convs["one_round_image"], "modal_inputs": {"vision": ["x.jpg"]}},
                                   {"conversations": convs["text_only"], "modal_inputs": {"video": ["v.mp4"]}}], tok)
    n0 = sum(len(c["value"].split()) for c in convs["text_only"])
    n1 = sum(len(c["value"].split()) for c in convs["one_round_image"])
    assert ds2.modality_lengths == [-n0, n1 + 256, n0 + 257 * 8]


def test_convert_merge_strategies_match_the_reference_script(tmp_path):
    """merge_unimodal_modelcompose.py:42-73 `convert-*`: 'same'-strategy checkpoints -> 'modal+language' + per-modality copies, then
    online-merge(-reset) / sum / mean on the result (the ties / drop forms run on the GPU: tests/test_merge_gpu.py)."""
    from conftest import run_g16_cases
    from modelcompose_amd import compose
    run_g16_cases(tmp_path, ["online", "online_plain", "sum", "mean"])
    with pytest.raises(NotImplementedError):
        compose.merge_checkpoints([str(tmp_path / "ckpt-vision"), str(tmp_path / "ckpt-audio")], str(tmp_path / "x"), "drop-mean")
    # a checkpoint that is not 'same'-strategy is refused, as the reference's assert does
    import json
    cfgp = tmp_path / "ckpt-audio" / "config.json"
    cfg = json.load(open(cfgp))
    cfg["lora_strategy"] = "modal+language"
    json.dump(cfg, open(cfgp, "w"))
    with pytest.raises(AssertionError):
        compose.merge_checkpoints([str(tmp_path / "ckpt-vision"), str(tmp_path / "ckpt-audio")], str(tmp_path / "y"), "convert-sum")


def test_parallel_synthetic_weights_equal_the_sequential_draw(monkeypatch):
    """MC_SYNTH_THREADS (the GPU test session sets it): the main thread only advances torch's CPU generator past every tensor, worker threads
    draw the tensors from the saved states - the state dict must be the sequential draw bit for bit (the committed full-depth fixtures were
    made from it), including tensors whose element count is not a multiple of 16 (the vectorised normal draws 16 extra values) and the
    kaiming-uniform LoRA factors."""
    import torch
    from modelcompose_amd import synthetic
    meta = synthetic.vicuna7b_meta(("vision", "audio", "point"), "default-vision=0.333,default-audio=0.333,default-point=0.333", layers=2)
    meta.update(vocab_size=1001, hidden_size=520, intermediate_size=1096, num_attention_heads=8, num_key_value_heads=8, lora_r=24)
    meta["clip"].update(hidden_size=264, intermediate_size=520, num_hidden_layers=2, image_size=70)
    meta["beats"].update(encoder_layers=2, encoder_embed_dim=264, encoder_ffn_embed_dim=520)
    meta["point"].update(depth=2)
    monkeypatch.delenv("MC_SYNTH_THREADS", raising=False)
    seq = synthetic.synthetic_state_dict(meta, device="cpu", seed=77, dtype=torch.bfloat16)
    monkeypatch.setenv("MC_SYNTH_THREADS", "4")
    par = synthetic.synthetic_state_dict(meta, device="cpu", seed=77, dtype=torch.bfloat16)
    assert list(seq) == list(par)
    big = [k for k, v in seq.items() if v.numel() >= 4096]
    assert any(seq[k].numel() % 16 for k in big) and any(".lora_A." in k for k in big)
    for k in seq:
        assert torch.equal(seq[k], par[k]), k

