"""TEST INFRASTRUCTURE — build-container only.

Generates the golden fixtures under tests/golden/ by running the UNMODIFIED
reference (/root/reference, imported through oracle/refshim.py) on tiny seeded
configurations.  Fixtures are data only: inputs, weights and the reference's
outputs as fp32/int arrays in .npz files (plus JSON for configs).  Re-run with

    python -m oracle.gen_golden            # all groups
    python -m oracle.gen_golden g1 g4      # some groups

Groups:
  g1  LocalLoraLinear per-adapter outputs + scaling dict
  g2  one decoder layer: routed prefill and cached decode
  g3  splice (equal-length and ragged batches, labels, masks)
  g4  tiny end-to-end: CLIP tower + mlp2x_gelu + splice + 2-layer LLM + greedy ids
  g5  encoders / projectors (clip, ...)
  g6  merge script file-level outputs
  g7  dense-merge equivalence
  g8  tiny 4-modality composed model end to end (all encoders + projectors + routed LLM + greedy ids)
  g9  stage-2 finetune step: loss + gradients of the trainable set
  g10 TIES merging (ties-mean / sum / max) tensor- and file-level
  g11 interference metrics (L2 / cosine / SSD / TSSD) tensor- and file-level
  g12 host callers: length-grouped samplers, LLaVA -> multimodal checkpoint conversion
  g13 prompt helpers of mm_utils.py: placeholder tokenisation, stopping criteria, expand2square
  g14 caller-side data formats: preprocess (conversation -> ids / labels) and the batch collator
  g15 FULL DEPTH: the metric's model (3-way composed Vicuna-7B, 32 layers, real widths, real-size encoders) on two rows of the metric's
      inputs, 17 greedy tokens, by the pinned oracle (oracle/pipeline.py, fp32 branch form).  NOT part of the default run: 45 GB of
      memory and ~15 minutes of CPU; `python -m oracle.gen_golden g15`.  The reference classes themselves cannot be instantiated at
      this size in 64 GB next to the oracle, so this group is oracle output (pinned by g1-g8), not reference output.
"""
from __future__ import annotations

import copy
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

from . import refshim

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    if isinstance(t, torch.Tensor):
        t = t.detach()
        if t.dtype == torch.bool:
            return t.numpy()
        if t.is_floating_point():
            return t.float().numpy()
        return t.numpy()
    return np.asarray(t)


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: _np(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def _sd(module, prefix="sd::", clip_prefix=None):
    """state_dict with the reference-era (transformers==4.31) key grammar: HF 5.x dropped the
    'vision_model.' wrapper of CLIPVisionModel; real checkpoints carry it, so fixtures do too."""
    out = {}
    for k, v in module.state_dict().items():
        if clip_prefix is not None and k.startswith(clip_prefix) and not k.startswith(clip_prefix + "vision_model."):
            k = clip_prefix + "vision_model." + k[len(clip_prefix):]
        out[prefix + k] = v
    return out


def _randomize_lora(model, seed=1):
    """reset_lora_parameters leaves B = 0; give every adapter a non-zero B so composition matters."""
    g = torch.Generator().manual_seed(seed)
    for n, p in model.named_parameters():
        if ".lora_B." in n:
            p.data = torch.randn(p.shape, generator=g) * 0.05
        if "prefix_tokens" in n or "suffix_tokens" in n:
            p.data = torch.randn(p.shape, generator=g) * 0.02


def tiny_llm_config(ml, modal=("vision", "audio"), reset="default-vision=0.5,default-audio=0.25", layers=2,
                    prefix_tokens=0, hidden=32, heads=4, inter=64, vocab=97, r=4, alpha=8):
    cfg = ml.MultimodalConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                              num_attention_heads=heads, num_key_value_heads=heads, max_position_embeddings=256,
                              rms_norm_eps=1e-5, pad_token_id=0, bos_token_id=1, eos_token_id=2)
    cfg.lora_r, cfg.lora_alpha, cfg.lora_dropout = r, alpha, 0.0
    cfg.lora_strategy = "modal+language"
    for m in modal:
        setattr(cfg, f"mm_{m}_encoder", f"fake/{m}")
    cfg.reset_scaling_weights = reset
    cfg.pretraining_tp = 1
    cfg.rope_scaling = None
    cfg.local_prefix_tokens = prefix_tokens
    cfg.local_suffix_tokens = prefix_tokens
    return cfg


def cfg_json(cfg, extra=None):
    keys = ["vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
            "num_key_value_heads", "max_position_embeddings", "rms_norm_eps", "lora_r", "lora_alpha", "lora_strategy",
            "reset_scaling_weights", "local_prefix_tokens", "local_suffix_tokens", "mm_vision_encoder",
            "mm_audio_encoder", "mm_video_encoder", "mm_point_encoder", "pad_token_id", "eos_token_id"]
    d = {k: getattr(cfg, k, None) for k in keys}
    d.update(extra or {})
    return json.dumps(d)


# ----------------------------------------------------------------------------
def g1():
    ml = refshim.import_ref("modelcompose.model.language_model.multimodal_llama")
    torch.manual_seed(11)
    names = ["default", "audio", "vision", "video"]
    lin = ml.LocalLoraLinear(names, 24, 40, 4, 8, 0.0, bias=False,
                             reset_scaling_weights="default-video=0.333,default-audio=0.5,default-vision=0.25,audio=1.5")
    lin.eval()
    _randomize_lora(lin, 3)
    x = torch.randn(2, 7, 24)
    outs = lin(x, active_adapters=names + ["point"])      # 'point' is unknown -> base output (:127-129)
    base = lin(x)
    arrays = {"x": x, "out::base": base}
    for k, v in outs.items():
        arrays[f"out::{k}"] = v
    arrays.update(_sd(lin))
    _save("g1_lora_linear", meta=np.array(json.dumps({
        "modal_names": names, "lora_r": 4, "lora_alpha": 8,
        "reset_scaling_weights": lin.reset_scaling_weights,
        "scaling": lin.scaling, "adapters": list(lin.lora_A.keys()),
        "default_adapter_names": lin.default_adapter_names, "merge_default_weights": lin.merge_default_weights})),
        **arrays)


def g2():
    ml = refshim.import_ref("modelcompose.model.language_model.multimodal_llama")
    torch.manual_seed(12)
    cfg = tiny_llm_config(ml, layers=2)
    model = ml.MultimodalLlamaForCausalLM(cfg).eval()
    _randomize_lora(model, 5)
    B, L = 2, 9
    x = torch.randn(B, L, cfg.hidden_size)
    mm = {"vision": torch.zeros(B, L, dtype=torch.bool), "audio": torch.zeros(B, L, dtype=torch.bool)}
    mm["vision"][0, 2:5] = True
    mm["vision"][1, 1:3] = True
    mm["audio"][0, 6:8] = True
    mm["default"] = (mm["vision"].int() + mm["audio"].int()) == 0
    am = torch.ones(B, L, dtype=torch.bool)
    am[1, -2:] = False       # right padding on sample 1
    with torch.no_grad():
        out = model.model(inputs_embeds=x, attention_mask=am, modal_attention_mask=mm, use_cache=True)
        h_pre, kv = out[0], out[1]
        # decode step with cache (modal mask dropped inside the layer, :435-438)
        x1 = torch.randn(B, 1, cfg.hidden_size)
        am1 = torch.ones(B, L + 1, dtype=torch.bool)
        out1 = model.model(inputs_embeds=x1, attention_mask=am1, past_key_values=kv, modal_attention_mask=mm,
                           use_cache=True)
        # unrouted prefill (modal mask None)
        out_nr = model.model(inputs_embeds=x, attention_mask=am, modal_attention_mask=None, use_cache=True)
        logits = model.lm_head(h_pre)
    arrays = dict(x=x, attention_mask=am, x1=x1, hidden_prefill=h_pre, hidden_decode=out1[0],
                  hidden_prefill_unrouted=out_nr[0], logits_prefill=logits,
                  k0=kv[0][0], v0=kv[0][1], k1_dec=out1[1][1][0])
    for k, v in mm.items():
        arrays[f"mask::{k}"] = v
    arrays.update(_sd(model))
    _save("g2_decoder", meta=np.array(cfg_json(cfg, {"modal_names": model.modal_names})), **arrays)


def g3():
    ml = refshim.import_ref("modelcompose.model.language_model.multimodal_llama")
    torch.manual_seed(13)
    cfg = tiny_llm_config(ml, modal=("vision", "audio", "video"), reset=None, layers=1, hidden=16, heads=2, inter=32)
    model = ml.MultimodalLlamaForCausalLM(cfg).eval()
    H = cfg.hidden_size

    class FakeEnc(nn.Module):
        def __init__(self, n, d):
            super().__init__()
            self.n, self.d = n, d
            self.p = nn.Parameter(torch.zeros(1))

        @property
        def dummy_inputs(self):
            return torch.zeros(1, self.n, self.d)

        def forward(self, x, **kw):
            return x

    class FakeAudio(FakeEnc):
        @property
        def dummy_inputs(self):
            return {"audio_inputs": torch.zeros(1, self.n, self.d)}

        def forward(self, audio_inputs, audio_padding_mask=None):
            return audio_inputs, None

    class FakeVideo(FakeEnc):
        @property
        def dummy_inputs(self):
            return torch.zeros(1, 2, self.n // 2, self.d)

        def forward(self, x):
            return x        # (b, t, n, d)

    model.model.modal_encoders = nn.ModuleDict({"audio": FakeAudio(3, H), "vision": FakeEnc(4, H), "video": FakeVideo(6, H)})
    model.model.modal_projectors = nn.ModuleDict({m: nn.Identity() for m in ("audio", "vision", "video")})
    pre = {m: torch.randn(1, 2, H) for m in ("default", "audio", "vision", "video")}
    suf = {m: torch.randn(1, 1, H) for m in ("default", "audio", "vision", "video")}

    def run(tag, ids, labels, am, modal_inputs, use_ps=True):
        with torch.no_grad():
            _, am2, _, emb, lab, mam = model.prepare_inputs_labels_for_multimodal(
                ids, am, None, labels, modal_inputs, pre if use_ps else None, suf if use_ps else None)
        arrays = {f"{tag}::input_ids": ids, f"{tag}::attention_mask_in": am, f"{tag}::embeds": emb,
                  f"{tag}::attention_mask": am2}
        if labels is not None:
            arrays[f"{tag}::labels_in"] = labels
            arrays[f"{tag}::labels"] = lab
        for k, v in mam.items():
            arrays[f"{tag}::mask::{k}"] = v
        for k, v in modal_inputs.items():
            if isinstance(v, dict):
                arrays[f"{tag}::modal::{k}"] = v["audio_inputs"]
            else:
                arrays[f"{tag}::modal::{k}"] = v
        return arrays

    arrays = {}
    V, A, VD = -200, -203, -204
    # (a) equal-length batch, vision+audio each, video absent (dummy path), with labels
    ids = torch.tensor([[1, 5, V, 13, 7, A, 13, 9, 10], [1, 6, A, 13, 8, V, 13, 11, 12]])
    labels = ids.clone(); labels[:, :4] = -100
    mi = {"vision": torch.randn(2, 4, H), "audio": {"audio_inputs": torch.randn(2, 3, H)}}
    arrays.update(run("eq", ids, labels, torch.ones_like(ids, dtype=torch.bool), mi))
    # (b) ragged: sample 0 has two images, sample 1 has video only; labels given (needed by the reference's ragged branch)
    ids = torch.tensor([[1, V, 13, V, 13, 20, 21], [1, 22, VD, 13, 23, 24, 25]])
    labels = ids.clone(); labels[:, :2] = -100
    am = torch.ones_like(ids, dtype=torch.bool)
    mi = {"vision": torch.randn(2, 4, H), "video": torch.randn(1, 2, 3, H)}
    # NB: audio absent here -> reference stacks an empty list for it in the ragged branch
    # (Appendix B); drop audio from the model for this case.
    enc_bak, proj_bak = model.model.modal_encoders, model.model.modal_projectors
    model.model.modal_encoders = nn.ModuleDict({"vision": enc_bak["vision"], "video": enc_bak["video"]})
    model.model.modal_projectors = nn.ModuleDict({"vision": nn.Identity(), "video": nn.Identity()})
    cfg.mm_audio_encoder = None
    arrays.update(run("ragged", ids, labels, am, mi))
    # (c) no prefix/suffix, no labels, equal length, sentinel first/last positions
    ids = torch.tensor([[V, 13, 30, 31, VD], [VD, 13, 32, 33, V]])
    mi = {"vision": torch.randn(2, 4, H), "video": torch.randn(2, 2, 3, H)}
    arrays.update(run("edge", ids, None, torch.ones_like(ids, dtype=torch.bool), mi, use_ps=False))
    model.model.modal_encoders, model.model.modal_projectors = enc_bak, proj_bak
    cfg.mm_audio_encoder = "fake/audio"
    for m in pre:
        arrays[f"prefix::{m}"] = pre[m]
        arrays[f"suffix::{m}"] = suf[m]
    arrays["embed_tokens"] = model.model.embed_tokens.weight
    _save("g3_splice", **arrays)


def _tiny_clip_dir(tmp, hidden=32, layers=3, heads=4, inter=64, image=28, patch=14, seed=21):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    torch.manual_seed(seed)
    c = CLIPVisionConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                         image_size=image, patch_size=patch, projection_dim=16)
    m = CLIPVisionModel(c).eval()
    for p in m.parameters():       # HF init leaves biases at 0 / LN at 1: randomise so every term matters
        p.data = torch.randn_like(p) * 0.05 + (1.0 if p.ndim == 1 and "norm" in "" else 0.0)
    for n, p in m.named_parameters():
        if "layer_norm" in n or "layrnorm" in n or "layernorm" in n:
            if n.endswith("weight"):
                p.data = 1.0 + 0.1 * torch.randn_like(p)
    d = os.path.join(tmp, "clip-tiny")
    m.save_pretrained(d, safe_serialization=False)
    json.dump({"crop_size": image, "do_center_crop": True, "do_normalize": True, "do_resize": True,
               "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711],
               "size": image, "image_processor_type": "CLIPImageProcessor"},
              open(os.path.join(d, "preprocessor_config.json"), "w"))
    return d, c


def _g4_build(tmp):
    """The tiny vision model and inputs of g4 (CLIPVisionTower -> mlp2x_gelu -> splice -> 2-layer LLM), built from the reference classes with
    fixed seeds: g4 and g18 call it and get the same weights and inputs."""
    ml = refshim.import_ref("modelcompose.model.language_model.multimodal_llama")
    ce = refshim.import_ref("modelcompose.model.multimodal_encoder.clip_encoder")
    pb = refshim.import_ref("modelcompose.model.multimodal_projector.builder")
    # dims the HIP path supports: head_dim 64, hidden % 64 == 0, vocab % 4 == 0
    clip_dir, ccfg = _tiny_clip_dir(tmp, hidden=128, layers=3, heads=2, inter=256)
    torch.manual_seed(14)
    cfg = tiny_llm_config(ml, modal=("vision",), reset="default-vision=0.5", layers=2, prefix_tokens=2,
                          hidden=128, heads=2, inter=192, vocab=128)
    cfg.mm_vision_encoder = clip_dir
    cfg.mm_vision_select_layer = -2
    cfg.mm_vision_select_feature = "patch"
    cfg.mm_projector_type = "mlp2x_gelu"
    model = ml.MultimodalLlamaForCausalLM(cfg).eval()
    args = types.SimpleNamespace(mm_vision_select_layer=-2, mm_vision_select_feature="patch")
    tower = ce.CLIPVisionTower(clip_dir, args, delay_load=False)
    pcfg = types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=ccfg.hidden_size, hidden_size=cfg.hidden_size)
    proj = pb.build_vision_projector(pcfg)
    for p in proj.parameters():
        p.data = torch.randn_like(p) * 0.1
    model.model.modal_encoders = nn.ModuleDict({"vision": tower})
    model.model.modal_projectors = nn.ModuleDict({"vision": proj})
    _randomize_lora(model, 7)
    model.eval()
    B = 2
    V = -200
    g = torch.Generator().manual_seed(3)
    ids = torch.cat([torch.ones(B, 1, dtype=torch.long), torch.randint(3, 97, (B, 4), generator=g),
                     torch.full((B, 1), V), torch.full((B, 1), 13), torch.randint(3, 97, (B, 5), generator=g)], dim=1)
    pixels = torch.randn(B, 3, 28, 28, generator=g)
    return model, tower, cfg, ccfg, ids, pixels


def g4():
    """Tiny end-to-end through the reference classes: CLIPVisionTower -> mlp2x_gelu -> splice -> 2-layer LLM."""
    with tempfile.TemporaryDirectory() as tmp:
        model, tower, cfg, ccfg, ids, pixels = _g4_build(tmp)
        B = ids.shape[0]
        n_new = 8
        with torch.no_grad():
            feats = tower(pixels)
            out = model(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool),
                        modal_inputs={"vision": pixels}, use_cache=True)
            logits0 = out.logits
            # hand-written 4.31-style greedy loop
            kv = out.past_key_values
            nxt = logits0[:, -1].argmax(-1)
            gen = [nxt]
            steps_logits = [logits0[:, -1]]
            am = torch.ones(B, logits0.shape[1], dtype=torch.bool)
            for _ in range(n_new - 1):
                am = torch.cat([am, torch.ones(B, 1, dtype=torch.bool)], dim=1)
                o = model(input_ids=nxt[:, None], attention_mask=am, past_key_values=kv,
                          modal_inputs={"vision": pixels}, use_cache=True)
                kv = o.past_key_values
                steps_logits.append(o.logits[:, -1])
                nxt = o.logits[:, -1].argmax(-1)
                gen.append(nxt)
        arrays = dict(input_ids=ids, pixels=pixels, clip_features=feats, logits_prefill=logits0,
                      gen_ids=torch.stack(gen, 1), step_logits=torch.stack(steps_logits, 1))
        arrays.update(_sd(model, clip_prefix="model.modal_encoders.vision.vision_tower."))
        extra = {"modal_names": model.modal_names, "mm_projector_type": "mlp2x_gelu", "mm_vision_select_layer": -2,
                 "clip": {"hidden_size": ccfg.hidden_size, "intermediate_size": ccfg.intermediate_size,
                          "num_hidden_layers": ccfg.num_hidden_layers, "num_attention_heads": ccfg.num_attention_heads,
                          "image_size": ccfg.image_size, "patch_size": ccfg.patch_size,
                          "layer_norm_eps": ccfg.layer_norm_eps, "hidden_act": ccfg.hidden_act}}
        d = json.loads(cfg_json(cfg, extra))
        d["mm_vision_encoder"] = "clip-tiny"
        _save("g4_e2e_vision", meta=np.array(json.dumps(d)), **arrays)


def g18():
    """forward(output_hidden_states=True, output_attentions=True) of the reference on g4's model and inputs (multimodal_llama.py:561-604,
    :295-312, :676-745): the tuple of hidden states (input of every layer, then the final norm) and every layer's attention probabilities.
    The weights are g4's (tests load them from g4_e2e_vision.npz); logits_prefill is stored again so that a test can tie the two fixtures."""
    with tempfile.TemporaryDirectory() as tmp:
        model, tower, cfg, ccfg, ids, pixels = _g4_build(tmp)
        with torch.no_grad():
            out = model(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool), modal_inputs={"vision": pixels},
                        output_hidden_states=True, output_attentions=True, return_dict=True)
        assert len(out.hidden_states) == cfg.num_hidden_layers + 1 and len(out.attentions) == cfg.num_hidden_layers
        _save("g18_hidden_attn", input_ids=ids, logits_prefill=out.logits, hidden_states=torch.stack(list(out.hidden_states), 0),
              attentions=torch.stack(list(out.attentions), 0))


def g6():
    """File-level golden for merge_checkpoints('online-merge-reset-...') (merge_unimodal_modelcompose.py:28-145)."""
    refshim.install()
    import importlib
    mm = importlib.import_module("merge_unimodal_modelcompose")
    torch.manual_seed(16)
    with tempfile.TemporaryDirectory() as tmp:
        paths = []
        inputs = {}
        for modal, enc_key in (("video", "mm_video_encoder"), ("audio", "mm_audio_encoder"), ("vision", "mm_vision_encoder")):
            d = os.path.join(tmp, f"ckpt-{modal}")
            os.makedirs(d)
            w = {}
            for i in range(2):
                for lin in ("self_attn.q_proj", "mlp.down_proj"):
                    for ab, shape in (("A", (4, 8)), ("B", (8, 4))):
                        w[f"model.layers.{i}.{lin}.lora_{ab}.default.weight"] = torch.randn(shape)
                        w[f"model.layers.{i}.{lin}.lora_{ab}.{modal}.weight"] = torch.randn(shape)
            w[f"model.modal_projectors.{modal}.0.weight"] = torch.randn(8, 6)
            w[f"prefix_tokens.{modal}"] = torch.randn(1, 2, 8)
            torch.save(w, os.path.join(d, "adapter_model.bin"))
            c = {"model_type": "multimodal", enc_key: f"/ckpts/{modal}", "lora_r": 4, "lora_alpha": 8,
                 "lora_strategy": "modal+language", "local_prefix_tokens": 2, "hidden_size": 8,
                 f"mm_{modal}_projector_type": "mlp2x_gelu"}
            json.dump(c, open(os.path.join(d, "config.json"), "w"))
            paths.append(d)
            inputs[modal] = (w, c)
        out = os.path.join(tmp, "merged")
        strat = "online-merge-reset-default-video=0.333,default-audio=0.333,default-vision=0.333"
        mm.merge_checkpoints(paths, out, strat)
        merged = torch.load(os.path.join(out, "adapter_model.bin"))
        mcfg = json.load(open(os.path.join(out, "config.json")))
        info = open(os.path.join(out, "merge_info.txt")).read().replace(tmp, "<TMP>")
        arrays = {}
        for modal, (w, c) in inputs.items():
            for k, v in w.items():
                arrays[f"in::{modal}::{k}"] = v
        for k, v in merged.items():
            arrays[f"out::{k}"] = v
        meta = {"strategy": strat, "order": ["video", "audio", "vision"],
                "in_configs": {m: c for m, (w, c) in inputs.items()}, "out_config": mcfg, "merge_info": info}
        _save("g6_merge", meta=np.array(json.dumps(meta)), **arrays)


def g7():
    """Dense-merge equivalence W + Σ c·(α/r)·B·A  vs branch form, through the reference layer."""
    ml = refshim.import_ref("modelcompose.model.language_model.multimodal_llama")
    torch.manual_seed(17)
    names = ["default", "audio", "vision"]
    lin = ml.LocalLoraLinear(names, 48, 32, 8, 16, 0.0, bias=False,
                             reset_scaling_weights="default-audio=0.333,default-vision=0.667").eval()
    _randomize_lora(lin, 9)
    x = torch.randn(3, 5, 48)
    outs = lin(x, active_adapters=names)
    arrays = {"x": x}
    for k, v in outs.items():
        arrays[f"out::{k}"] = v
    arrays.update(_sd(lin))
    _save("g7_dense_merge", meta=np.array(json.dumps({"modal_names": names, "lora_r": 8, "lora_alpha": 16,
                                                      "reset_scaling_weights": lin.reset_scaling_weights,
                                                      "scaling": lin.scaling})), **arrays)


def g5_clip():
    ce = refshim.import_ref("modelcompose.model.multimodal_encoder.clip_encoder")
    with tempfile.TemporaryDirectory() as tmp:
        clip_dir, ccfg = _tiny_clip_dir(tmp, hidden=128, layers=3, heads=2, inter=256, image=42, patch=14, seed=31)
        args = types.SimpleNamespace(mm_vision_select_layer=-2, mm_vision_select_feature="patch")
        tower = ce.CLIPVisionTower(clip_dir, args, delay_load=False).eval()
        torch.manual_seed(5)
        px = torch.randn(2, 3, 42, 42)
        with torch.no_grad():
            f = tower(px)
            hs = tower.vision_tower(px, output_hidden_states=True).hidden_states
            tower.select_feature = "cls_patch"; tower.select_layer = -1
            f_last = tower(px)
        arrays = dict(pixels=px, features=f, features_last_cls=f_last, hs0=hs[0], hs1=hs[1])
        arrays.update(_sd(tower.vision_tower, clip_prefix=""))
        meta = {"hidden_size": 128, "intermediate_size": 256, "num_hidden_layers": 3, "num_attention_heads": 2,
                "image_size": 42, "patch_size": 14, "layer_norm_eps": ccfg.layer_norm_eps, "hidden_act": ccfg.hidden_act}
        _save("g5_clip", meta=np.array(json.dumps(meta)), **arrays)


def _jitter(module, seed, scale=0.05):
    """Randomise every parameter/buffer a little so that no term is trivially 0/1."""
    g = torch.Generator().manual_seed(seed)
    for n, p_ in list(module.named_parameters()) + [(n, b) for n, b in module.named_buffers() if b.is_floating_point()]:
        if "running_var" in n:
            p_.data = 0.5 + torch.rand(p_.shape, generator=g)
        elif p_.ndim <= 1 and ("norm" in n.lower() or "ln" in n.lower()) and n.endswith("weight"):
            p_.data = 1.0 + 0.1 * torch.randn(p_.shape, generator=g)
        else:
            p_.data = p_.data + scale * torch.randn(p_.shape, generator=g)


def g5_beats():
    bm = refshim.import_ref("modelcompose.model.multimodal_encoder.beats.BEATs")
    cfgd = dict(input_patch_size=16, embed_dim=64, conv_bias=False, encoder_layers=2, encoder_embed_dim=128,
                encoder_ffn_embed_dim=256, encoder_attention_heads=2, activation_fn="gelu", layer_norm_first=False, deep_norm=True,
                dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0,
                conv_pos=8, conv_pos_groups=2, relative_position_embedding=True, num_buckets=32, max_distance=64, gru_rel_pos=True,
                finetuned_model=False)
    torch.manual_seed(41)
    model = bm.BEATs(bm.BEATsConfig(cfgd)).eval()
    _jitter(model, 42)
    fbank = torch.randn(2, 64, 128) * 0.5
    pad = torch.zeros(2, 64, dtype=torch.bool)
    pad[1, 40:] = True                      # trailing padding on sample 1 (32 of 64 frames -> tokens 20.. padded)
    fbank[1, 40:] = 0
    with torch.no_grad():
        f, pm = model.extract_features_new(fbank.clone(), pad.clone(), feature_only=True)
        f_nopad, _ = model.extract_features_new(fbank.clone(), None, feature_only=True)
    arrays = dict(fbank=fbank, padding_mask=pad, features=f, pooled_mask=pm, features_nopad=f_nopad)
    arrays.update(_sd(model))
    _save("g5_beats", meta=np.array(json.dumps(cfgd)), **arrays)


def g5_qformer():
    pb = refshim.import_ref("modelcompose.model.multimodal_projector.builder")
    real = pb.BertConfig
    small = dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, vocab_size=64, max_position_embeddings=16)
    pb.BertConfig = lambda **kw: real(**{**small, **kw})
    import transformers
    iw = transformers.PreTrainedModel.init_weights
    # transformers==4.31 init_weights = apply(_init_weights) (+ tying, nothing is tied once word embeddings are dropped)
    transformers.PreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    try:
        torch.manual_seed(43)
        proj = pb.VideoLlamaAudioQformer(num_query_token=4, vision_width=128, num_hidden_layers=2, num_positions=64).eval()
    finally:
        pb.BertConfig = real
        transformers.PreTrainedModel.init_weights = iw
    _jitter(proj, 44)
    x = torch.randn(2, 32, 128)
    # ModuleUtilsMixin.get_head_mask of transformers==4.31 with head_mask=None: [None] * num_hidden_layers
    if not hasattr(transformers.PreTrainedModel, "get_head_mask"):
        transformers.PreTrainedModel.get_head_mask = lambda self, head_mask, n, is_attention_chunked=False: [None] * n
    with torch.no_grad():
        y = proj(x)
    arrays = dict(x=x, y=y)
    arrays.update(_sd(proj))
    meta = dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2, layer_norm_eps=1e-12,
                num_query_token=4, encoder_width=128, out_features=y.shape[-1])
    _save("g5_qformer", meta=np.array(json.dumps(meta)), **arrays)


def g5_video():
    mv = refshim.import_ref("modelcompose.model.multimodal_encoder.languagebind.video.modeling_video")
    cv = refshim.import_ref("modelcompose.model.multimodal_encoder.languagebind.video.configuration_video")
    vc = cv.CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=28,
                             patch_size=14, num_frames=2, add_time_attn=True, force_patch_dropout=0.0)
    torch.manual_seed(45)
    tower = mv.CLIPVisionTransformer(vc).eval()
    _jitter(tower, 46)
    video = torch.randn(2, 3, 2, 28, 28)
    with torch.no_grad():
        out = tower(video, output_hidden_states=True, return_dict=True)
    hs = out.hidden_states
    arrays = dict(video=video, hs_m2=hs[-2], hs0=hs[0], hs1=hs[1])
    arrays.update(_sd(tower))
    meta = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=28, patch_size=14,
                num_frames=2, add_time_attn=True, layer_norm_eps=vc.layer_norm_eps, hidden_act=vc.hidden_act)
    _save("g5_video", meta=np.array(json.dumps(meta)), **arrays)


def g5_point():
    pe = refshim.import_ref("modelcompose.model.multimodal_encoder.pointbert.point_encoder")
    from easydict import EasyDict
    cfgd = dict(trans_dim=128, depth=2, drop_path_rate=0.0, cls_dim=40, num_heads=2, group_size=8, num_group=16, encoder_dims=64,
                point_dims=6, use_max_pool=False)
    torch.manual_seed(47)
    model = pe.PointTransformer(EasyDict(cfgd), use_max_pool=False)
    # SyncBatchNorm refuses CPU tensors; in eval mode it is the running-stat affine of BatchNorm1d (same state_dict keys)
    for seq in (model.encoder.first_conv, model.encoder.second_conv):
        bn = seq[1]
        nb = nn.BatchNorm1d(bn.num_features)
        nb.load_state_dict(bn.state_dict())
        seq[1] = nb
    model.eval()
    _jitter(model, 48)
    B, N = 2, 256
    g = torch.Generator().manual_seed(49)
    xyz = torch.randn(B, N, 3, generator=g)
    xyz = xyz / xyz.norm(dim=-1, keepdim=True).clamp_min(1e-6) * torch.rand(B, N, 1, generator=g) ** (1 / 3)
    pts = torch.cat([xyz, torch.rand(B, N, 3, generator=g)], -1)
    torch.manual_seed(50)
    start = torch.randint(0, N, (B,), dtype=torch.long)        # what misc.fps draws first (pointbert/misc.py:52)
    torch.manual_seed(50)
    with torch.no_grad():
        y = model(pts)
        torch.manual_seed(50)
        nbh, center = model.group_divider(pts)
    arrays = dict(points=pts, fps_start=start, features=y, center=center, neighborhood=nbh)
    arrays.update(_sd(model))
    _save("g5_point", meta=np.array(json.dumps(cfgd)), **arrays)


def g8():
    """Tiny 4-modality composed model end to end through the reference classes (BASELINE configs 3/4 in miniature):
    CLIP + mlp2x_gelu, BEATs + Q-Former, LanguageBind-Video + mlp2x_gelu, PointBERT + mlp2x_gelu, prefix/suffix tokens,
    4-way online-merge-reset coefficients, routed LocalLoRA prefill, cached greedy decode.  The encoder wrapper objects
    are the reference's own classes (their forward() is what runs); only their checkpoint-file loading is bypassed."""
    ml = refshim.import_ref("modelcompose.model.language_model.multimodal_llama")
    ce = refshim.import_ref("modelcompose.model.multimodal_encoder.clip_encoder")
    ae = refshim.import_ref("modelcompose.model.multimodal_encoder.audio_encoder")
    bm = refshim.import_ref("modelcompose.model.multimodal_encoder.beats.BEATs")
    lb = refshim.import_ref("modelcompose.model.multimodal_encoder.languagebind")
    mv = refshim.import_ref("modelcompose.model.multimodal_encoder.languagebind.video.modeling_video")
    cv = refshim.import_ref("modelcompose.model.multimodal_encoder.languagebind.video.configuration_video")
    pw = refshim.import_ref("modelcompose.model.multimodal_encoder.point_encoder")
    pe = refshim.import_ref("modelcompose.model.multimodal_encoder.pointbert.point_encoder")
    pb = refshim.import_ref("modelcompose.model.multimodal_projector.builder")
    from easydict import EasyDict
    import transformers
    H = 128
    with tempfile.TemporaryDirectory() as tmp:
        clip_dir, ccfg = _tiny_clip_dir(tmp, hidden=128, layers=3, heads=2, inter=256)
        torch.manual_seed(81)
        reset = "default-vision=0.25,default-audio=0.25,default-video=0.25,default-point=0.25"
        cfg = tiny_llm_config(ml, modal=("vision", "audio", "video", "point"), reset=reset, layers=2, prefix_tokens=2,
                              hidden=H, heads=2, inter=192, vocab=128)
        cfg.mm_vision_encoder = clip_dir
        cfg.mm_vision_select_layer, cfg.mm_vision_select_feature, cfg.mm_projector_type = -2, "patch", "mlp2x_gelu"
        model = ml.MultimodalLlamaForCausalLM(cfg).eval()
        # --- vision
        args = types.SimpleNamespace(mm_vision_select_layer=-2, mm_vision_select_feature="patch", mm_video_select_layer=-2)
        vis = ce.CLIPVisionTower(clip_dir, args, delay_load=False)
        # --- audio: BEATs inside the reference's BeatsAudioEncoder wrapper
        beats_cfg = dict(input_patch_size=16, embed_dim=64, conv_bias=False, encoder_layers=2, encoder_embed_dim=128,
                         encoder_ffn_embed_dim=256, encoder_attention_heads=2, activation_fn="gelu", layer_norm_first=False,
                         deep_norm=True, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0,
                         dropout_input=0.0, conv_pos=8, conv_pos_groups=2, relative_position_embedding=True, num_buckets=32,
                         max_distance=64, gru_rel_pos=True, finetuned_model=False)
        aud = ae.BeatsAudioEncoder.__new__(ae.BeatsAudioEncoder)
        nn.Module.__init__(aud)
        aud.cfg_only = bm.BEATsConfig(beats_cfg)
        aud.audio_encoder = bm.BEATs(aud.cfg_only).eval()
        aud.is_loaded = True
        _jitter(aud, 82)
        # --- video: LanguageBind CLIPVisionTransformer inside LanguageBindVideoTower
        vcd = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=28, patch_size=14,
                   num_frames=2, add_time_attn=True, force_patch_dropout=0.0)
        vc = cv.CLIPVisionConfig(**vcd)
        vid = lb.LanguageBindVideoTower.__new__(lb.LanguageBindVideoTower)
        nn.Module.__init__(vid)
        vid.select_layer, vid.select_feature, vid.is_loaded = -2, "patch", True
        vid.video_tower = mv.CLIPVisionTransformer(vc).eval()
        _jitter(vid, 83)
        # --- point: PointTransformer inside PointEncoder
        pcd = dict(trans_dim=128, depth=2, drop_path_rate=0.0, cls_dim=40, num_heads=2, group_size=8, num_group=16, encoder_dims=64,
                   point_dims=6, use_max_pool=False)
        pnt = pw.PointEncoder.__new__(pw.PointEncoder)
        nn.Module.__init__(pnt)
        pnt.cfg_only, pnt.is_loaded = EasyDict(pcd), True
        ptm = pe.PointTransformer(EasyDict(pcd), use_max_pool=False)
        for seq in (ptm.encoder.first_conv, ptm.encoder.second_conv):
            bn = seq[1]
            nb_ = nn.BatchNorm1d(bn.num_features)
            nb_.load_state_dict(bn.state_dict())
            seq[1] = nb_
        pnt.point_encoder = ptm.eval()
        _jitter(pnt, 84)
        # --- projectors (reference builders; the Q-Former's output Linear is hard-wired to 4096 -> resized to the tiny hidden)
        def mlp(in_f):
            pc = types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=in_f, hidden_size=H)
            m = pb.build_vision_projector(pc)
            return m
        real = pb.BertConfig
        small = dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, vocab_size=64, max_position_embeddings=16)
        pb.BertConfig = lambda **kw: real(**{**small, **kw})
        iw = transformers.PreTrainedModel.init_weights
        transformers.PreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
        try:
            torch.manual_seed(85)
            qf = pb.VideoLlamaAudioQformer(num_query_token=4, vision_width=128, num_hidden_layers=2, num_positions=64).eval()
        finally:
            pb.BertConfig = real
            transformers.PreTrainedModel.init_weights = iw
        qf.audio_llama_proj = nn.Linear(128, H)
        if not hasattr(transformers.PreTrainedModel, "get_head_mask"):
            transformers.PreTrainedModel.get_head_mask = lambda self, head_mask, n, is_attention_chunked=False: [None] * n
        projs = {"vision": mlp(128), "audio": qf, "video": mlp(128), "point": mlp(128)}
        for i, (k, m) in enumerate(projs.items()):
            _jitter(m, 86 + i, scale=0.08)
        model.model.modal_encoders = nn.ModuleDict({"vision": vis, "audio": aud, "video": vid, "point": pnt})
        model.model.modal_projectors = nn.ModuleDict(projs)
        _randomize_lora(model, 9)
        model.eval()
        # --- inputs: both samples carry all four modalities, in different orders (equal spliced length: the reference's
        # ragged branch needs labels, multimodal_arch.py:414-431, and its position ids ignore padding)
        g = torch.Generator().manual_seed(90)
        V, A, VD, P = -200, -203, -204, -205
        r = lambda n: torch.randint(3, 97, (n,), generator=g).tolist()
        s0 = [1] + r(3) + [V, 13] + [A, 13] + [VD, 13] + [P, 13] + r(4)
        s1 = [1] + r(2) + [P, 13] + r(1) + [VD, 13] + [A, 13] + r(2) + [V, 13] + r(2)
        assert len(s0) == len(s1)
        ids = torch.tensor([s0, s1], dtype=torch.long)
        pixels = torch.randn(2, 3, 28, 28, generator=g)
        fbank = torch.randn(2, 64, 128, generator=g) * 0.5
        pad = torch.zeros(2, 64, dtype=torch.bool)
        video = torch.randn(2, 3, 2, 28, 28, generator=g)
        N = 256
        xyz = torch.randn(2, N, 3, generator=g)
        xyz = xyz / xyz.norm(dim=-1, keepdim=True).clamp_min(1e-6) * torch.rand(2, N, 1, generator=g) ** (1 / 3)
        pts = torch.cat([xyz, torch.rand(2, N, 3, generator=g)], -1)
        torch.manual_seed(91)
        fps_start = torch.randint(0, N, (2,), dtype=torch.long)            # the draw misc.fps makes first (pointbert/misc.py:52)
        # modal_inputs dict order = collator order of first appearance (multimodal_dataset.py:171-173)
        mi = {"vision": pixels, "audio": {"audio_inputs": fbank, "audio_padding_mask": pad}, "video": video, "point": pts}
        n_new = 8

        def fwd(**kw):
            torch.manual_seed(91)                                          # FPS start is redrawn at every encode
            return model(modal_inputs=mi, use_cache=True, **kw)

        with torch.no_grad():
            torch.manual_seed(91)
            feats, _ = model.encode_modal_inputs(mi, model.prefix_tokens, model.suffix_tokens)
            out = fwd(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool))
            logits0 = out.logits
            kv = out.past_key_values
            nxt = logits0[:, -1].argmax(-1)
            gen, steps_logits = [nxt], [logits0[:, -1]]
            am = torch.ones(2, logits0.shape[1], dtype=torch.bool)
            for _ in range(n_new - 1):
                am = torch.cat([am, torch.ones(2, 1, dtype=torch.bool)], dim=1)
                o = fwd(input_ids=nxt[:, None], attention_mask=am, past_key_values=kv)
                kv = o.past_key_values
                steps_logits.append(o.logits[:, -1])
                nxt = o.logits[:, -1].argmax(-1)
                gen.append(nxt)
        arrays = dict(input_ids=ids, pixels=pixels, fbank=fbank, padding_mask=pad, video=video, points=pts, fps_start=fps_start,
                      logits_prefill=logits0, gen_ids=torch.stack(gen, 1), step_logits=torch.stack(steps_logits, 1))
        for m_, f_ in feats.items():
            arrays[f"feat_{m_}"] = f_
        arrays.update(_sd(model, clip_prefix="model.modal_encoders.vision.vision_tower."))
        extra = {"modal_names": model.modal_names, "mm_projector_type": "mlp2x_gelu", "mm_vision_select_layer": -2,
                 "mm_audio_projector_type": "qformer_4N_2L", "mm_video_projector_type": "mlp2x_gelu",
                 "mm_point_projector_type": "mlp2x_gelu", "mm_video_select_layer": -2, "fps_start": fps_start.tolist(),
                 "mm_audio_hidden_size": 128, "mm_video_hidden_size": 128, "mm_point_hidden_size": 128, "mm_hidden_size": 128,
                 "clip": {"hidden_size": ccfg.hidden_size, "intermediate_size": ccfg.intermediate_size,
                          "num_hidden_layers": ccfg.num_hidden_layers, "num_attention_heads": ccfg.num_attention_heads,
                          "image_size": ccfg.image_size, "patch_size": ccfg.patch_size,
                          "layer_norm_eps": ccfg.layer_norm_eps, "hidden_act": ccfg.hidden_act},
                 "beats": beats_cfg,
                 "video": dict(vcd, layer_norm_eps=vc.layer_norm_eps, hidden_act=vc.hidden_act),
                 "point": pcd,
                 "qformer": dict(hidden_size=128, num_attention_heads=2, intermediate_size=256, num_hidden_layers=2,
                                 layer_norm_eps=1e-12, num_query_token=4, encoder_width=128, out_features=H, num_positions=64)}
        d = json.loads(cfg_json(cfg, extra))
        d["mm_vision_encoder"] = "clip-tiny"
        _save("g8_e2e_4modal", meta=np.array(json.dumps(d)), **arrays)


def g5_imagebind():
    """ImageBind audio branch (the 'VideoLLaMA' audio encoder, multimodal_encoder/builder.py:91-95): reference ImageBindModel with
    tiny widths for every modality, only the audio preprocessor / trunk / head / postprocessor are exercised and stored."""
    im = refshim.import_ref("modelcompose.model.multimodal_encoder.imagebind.imagebind_model")
    torch.manual_seed(61)
    kw = dict(video_frames=2, kernel_size=(2, 14, 14), audio_kernel_size=16, audio_stride=10, out_embed_dim=64,
              vision_embed_dim=32, vision_num_blocks=1, vision_num_heads=2, audio_embed_dim=128, audio_num_blocks=2, audio_num_heads=2,
              audio_num_mel_bins=32, audio_target_len=46, audio_drop_path=0.1, text_embed_dim=32, text_num_blocks=1, text_num_heads=2,
              depth_embed_dim=32, depth_num_blocks=1, depth_num_heads=2, thermal_embed_dim=32, thermal_num_blocks=1,
              thermal_num_heads=2, imu_embed_dim=32, imu_num_blocks=1, imu_num_heads=2)
    model = im.ImageBindModel(**kw).eval()
    _jitter(model, 62)
    x = torch.randn(2, 3, 1, 32, 46)                       # (B, clips, 1, mel bins, frames)
    with torch.no_grad():
        feat, out = model.get_audio_feature(x.clone(), im.ModalityType.AUDIO)
        fwd = model(x.clone())
    assert torch.equal(fwd, out)
    arrays = dict(x=x, cls_feature=feat, out=out)
    keep = ("modality_preprocessors.audio.", "modality_trunks.audio.", "modality_heads.audio.", "modality_postprocessors.audio.")
    arrays.update({k: v for k, v in _sd(model).items() if k[4:].startswith(keep)})
    meta = dict(audio_kernel_size=16, audio_stride=10, audio_embed_dim=128, audio_num_blocks=2, audio_num_heads=2, audio_num_mel_bins=32,
                audio_target_len=46, out_embed_dim=64)
    _save("g5_imagebind", meta=np.array(json.dumps(meta)), **arrays)


def g9():
    """Stage-2 finetune step in miniature (BASELINE config 5): vision-only LocalLoRA model (adapters default + vision, no reset
    string), labels with IGNORE_INDEX over the prompt, loss = shifted cross-entropy (multimodal_llama.py:722-733) and its gradients
    w.r.t. the trainable set of train_multimodal.py:436-465 with lora_strategy='modal+language': every lora_A/lora_B, the
    mm_projector, prefix/suffix tokens.  lora_dropout = 0 so the step is deterministic."""
    ml = refshim.import_ref("modelcompose.model.language_model.multimodal_llama")
    ce = refshim.import_ref("modelcompose.model.multimodal_encoder.clip_encoder")
    pb = refshim.import_ref("modelcompose.model.multimodal_projector.builder")
    with tempfile.TemporaryDirectory() as tmp:
        clip_dir, ccfg = _tiny_clip_dir(tmp, hidden=128, layers=3, heads=2, inter=256)
        torch.manual_seed(94)
        cfg = tiny_llm_config(ml, modal=("vision",), reset=None, layers=2, prefix_tokens=2, hidden=128, heads=2, inter=192, vocab=128,
                              r=32, alpha=64)
        cfg.mm_vision_encoder = clip_dir
        cfg.mm_vision_select_layer, cfg.mm_vision_select_feature, cfg.mm_projector_type = -2, "patch", "mlp2x_gelu"
        model = ml.MultimodalLlamaForCausalLM(cfg).eval()
        args = types.SimpleNamespace(mm_vision_select_layer=-2, mm_vision_select_feature="patch")
        tower = ce.CLIPVisionTower(clip_dir, args, delay_load=False)
        pcfg = types.SimpleNamespace(mm_projector_type="mlp2x_gelu", mm_hidden_size=ccfg.hidden_size, hidden_size=cfg.hidden_size)
        proj = pb.build_vision_projector(pcfg)
        for p_ in proj.parameters():
            p_.data = torch.randn_like(p_) * 0.1
        model.model.modal_encoders = nn.ModuleDict({"vision": tower})
        model.model.modal_projectors = nn.ModuleDict({"vision": proj})
        _randomize_lora(model, 11)
        # trainable set (train_multimodal.py:436-465, lora_strategy='modal+language')
        model.requires_grad_(False)
        train = {}
        for n, p_ in model.named_parameters():
            if "prefix_tokens" in n or "suffix_tokens" in n or "lora" in n or n.startswith("model.modal_projectors."):
                p_.requires_grad = True
                train[n] = p_
        B, V = 3, -200
        g = torch.Generator().manual_seed(95)
        ids = torch.cat([torch.ones(B, 1, dtype=torch.long), torch.randint(3, 97, (B, 3), generator=g), torch.full((B, 1), V),
                         torch.full((B, 1), 13), torch.randint(3, 97, (B, 9), generator=g)], dim=1)
        labels = ids.clone()
        labels[:, :8] = -100                      # prompt (incl. the image sentinel) is not a target
        labels[1, 12:] = -100                     # ragged answer lengths
        pixels = torch.randn(B, 3, 28, 28, generator=g)
        out = model(input_ids=ids, attention_mask=torch.ones_like(ids, dtype=torch.bool), labels=labels, modal_inputs={"vision": pixels})
        out.loss.backward()
        arrays = dict(input_ids=ids, labels=labels, pixels=pixels, loss=out.loss.detach(), logits=out.logits.detach())
        for n, p_ in train.items():
            if p_.grad is None:                  # e.g. prefix_tokens.default: no text-modality block is ever spliced
                assert n.endswith("_tokens.default"), n
                continue
            arrays["grad::" + n] = p_.grad
        arrays.update(_sd(model, clip_prefix="model.modal_encoders.vision.vision_tower."))
        extra = {"modal_names": model.modal_names, "mm_projector_type": "mlp2x_gelu", "mm_vision_select_layer": -2,
                 "clip": {"hidden_size": ccfg.hidden_size, "intermediate_size": ccfg.intermediate_size,
                          "num_hidden_layers": ccfg.num_hidden_layers, "num_attention_heads": ccfg.num_attention_heads,
                          "image_size": ccfg.image_size, "patch_size": ccfg.patch_size,
                          "layer_norm_eps": ccfg.layer_norm_eps, "hidden_act": ccfg.hidden_act}}
        d = json.loads(cfg_json(cfg, extra))
        d["mm_vision_encoder"] = "clip-tiny"
        _save("g9_train_step", meta=np.array(json.dumps(d)), **arrays)


def g10():
    """TIES merging (scripts/model_composition/ties_merging.py:88-221, reached through merge_unimodal_modelcompose.py:78-93 with
    --strategy ties-{mean,sum,max}): the reference's own do_merging on three small fp32 checkpoints whose shared tensors contain
    magnitude ties at the trim threshold, columns whose trimmed sum is exactly zero (resolve_zero_signs) and exact zeros; plus
    the reference's demo() inputs (:253-256) and one file-level merge_checkpoints run."""
    refshim.install()
    import importlib
    mm = importlib.import_module("merge_unimodal_modelcompose")
    tm = importlib.import_module("ties_merging")
    g = torch.Generator().manual_seed(101)
    shapes = {"model.layers.0.self_attn.q_proj.lora_A.default.weight": (4, 24), "model.layers.0.self_attn.q_proj.lora_B.default.weight": (24, 4),
              "model.layers.1.mlp.down_proj.lora_A.default.weight": (4, 40), "model.layers.1.mlp.down_proj.lora_B.default.weight": (24, 4)}
    cks = []
    for i in range(3):
        w = {k: torch.randn(shp, generator=g) for k, shp in shapes.items()}
        cks.append(w)
    # engineered cases inside the first tensor (96 values): ties, cancelling columns, zeros
    k0 = "model.layers.0.self_attn.q_proj.lora_A.default.weight"
    for i in range(3):
        f = cks[i][k0].view(-1)
        f[0:4] = torch.tensor([2.5, -2.5, 2.5, 0.0])          # equal magnitudes (ties at / near the threshold)
        f[10] = 0.0
    cks[0][k0].view(-1)[5], cks[1][k0].view(-1)[5], cks[2][k0].view(-1)[5] = 3.0, -3.0, 0.0        # trimmed sum exactly 0
    cks[0][k0].view(-1)[6], cks[1][k0].view(-1)[6], cks[2][k0].view(-1)[6] = -4.0, 4.0, 0.0
    arrays = {}
    for i, w in enumerate(cks):
        for k, v in w.items():
            arrays[f"in::{i}::{k}"] = v
    import io, contextlib
    for func in ("mean", "sum", "max"):
        for K in (20, 50):
            with contextlib.redirect_stdout(io.StringIO()):
                out = tm.do_merging([dict(c) for c in cks], K=K, merge_func=f"dis-{func}")
            for k, v in out.items():
                arrays[f"out::{func}::{K}::{k}"] = v
    with contextlib.redirect_stdout(io.StringIO()):
        demo = tm.do_merging([{"x": torch.Tensor([1, 2, 3]), "y": torch.Tensor([4, 5, 6])},
                              {"x": torch.Tensor([-1, 2, 3]), "y": torch.Tensor([0, 0, 0])}], K=0.9)
    arrays["demo::x"], arrays["demo::y"] = demo["x"], demo["y"]
    # file level: three unimodal checkpoints (shared default keys + unique modal keys)
    with tempfile.TemporaryDirectory() as tmp:
        paths, in_cfg = [], {}
        for i, (modal, enc_key) in enumerate((("vision", "mm_vision_encoder"), ("audio", "mm_audio_encoder"), ("video", "mm_video_encoder"))):
            d = os.path.join(tmp, f"ckpt-{modal}")
            os.makedirs(d)
            w = dict(cks[i])
            w[f"model.layers.0.self_attn.q_proj.lora_A.{modal}.weight"] = torch.randn(4, 24, generator=g)
            w[f"model.modal_projectors.{modal}.0.weight"] = torch.randn(8, 6, generator=g)
            torch.save(w, os.path.join(d, "adapter_model.bin"))
            c = {"model_type": "multimodal", enc_key: f"/ckpts/{modal}", "lora_r": 4, "lora_alpha": 8, "lora_strategy": "modal+language"}
            json.dump(c, open(os.path.join(d, "config.json"), "w"))
            paths.append(d)
            in_cfg[modal] = c
            for k, v in w.items():
                if k not in cks[i]:
                    arrays[f"fin::{modal}::{k}"] = v
        outp = os.path.join(tmp, "merged")
        with contextlib.redirect_stdout(io.StringIO()):
            mm.merge_checkpoints(paths, outp, "ties-mean", K=20)
        merged = torch.load(os.path.join(outp, "adapter_model.bin"))
        for k, v in merged.items():
            arrays[f"fout::{k}"] = v
        meta = {"order": ["vision", "audio", "video"], "in_configs": in_cfg, "out_config": json.load(open(os.path.join(outp, "config.json"))),
                "merge_info": open(os.path.join(outp, "merge_info.txt")).read().replace(tmp, "<TMP>"), "shared_keys": sorted(shapes)}
    _save("g10_ties", meta=np.array(json.dumps(meta)), **arrays)


def g11():
    """Interference metrics (scripts/model_composition/calculate_metrics.py:26-37, :41-76): the reference's own L2 / cos_sim /
    soft_sign_dissimilarity / topk_values_mask on stacked task vectors (2 and 3 rows, with exact zeros, all-zero columns and
    cancelling columns), and one file-level calculate_metrics run on a ties-merged checkpoint directory."""
    refshim.install()
    import importlib, io, contextlib
    mm = importlib.import_module("merge_unimodal_modelcompose")
    tm = importlib.import_module("ties_merging")
    cm = importlib.import_module("calculate_metrics")
    g = torch.Generator().manual_seed(111)
    arrays, meta = {}, {"cases": []}
    for n, d in ((2, 4099), (3, 10000)):
        flat = torch.randn(n, d, generator=g) * 0.02
        flat[:, 5:40] = 0                                    # all-zero columns (excluded from the SSD mean)
        flat[0, 100:120] = 0                                 # single zeros
        flat[1, 200:210] = -flat[0, 200:210]                 # cancelling pairs
        if n == 3:
            flat[2, 200:210] = 0
        trunc, *_ = tm.topk_values_mask(flat.clone(), K=50, return_mask=False)
        exp = {"L2": float(cm.L2(flat)), "Cosine": float(cm.cos_sim(flat)), "SSD": float(cm.soft_sign_dissimilarity(flat)),
               "TSSD": float(cm.soft_sign_dissimilarity(trunc))}
        arrays[f"flat::{n}"] = flat
        meta["cases"].append({"n": n, "d": d, "expected": exp})
    with tempfile.TemporaryDirectory() as tmp:
        shapes = {"model.layers.0.self_attn.q_proj.lora_A.default.weight": (4, 24), "model.layers.0.self_attn.q_proj.lora_B.default.weight": (24, 4),
                  "model.layers.1.mlp.down_proj.lora_A.default.weight": (4, 40)}
        paths, in_cfg = [], {}
        for i, (modal, enc_key) in enumerate((("vision", "mm_vision_encoder"), ("audio", "mm_audio_encoder"))):
            dd = os.path.join(tmp, f"ckpt-{modal}")
            os.makedirs(dd)
            w = {k: torch.randn(shp, generator=g) for k, shp in shapes.items()}
            w[f"model.modal_projectors.{modal}.0.weight"] = torch.randn(8, 6, generator=g)
            torch.save(w, os.path.join(dd, "adapter_model.bin"))
            c = {"model_type": "multimodal", enc_key: f"/ckpts/{modal}", "lora_r": 4, "lora_alpha": 8, "lora_strategy": "modal+language"}
            json.dump(c, open(os.path.join(dd, "config.json"), "w"))
            paths.append(dd)
            in_cfg[modal] = c
            for k, v in w.items():
                arrays[f"fin::{modal}::{k}"] = v
        outp = os.path.join(tmp, "merged")
        with contextlib.redirect_stdout(io.StringIO()):
            mm.merge_checkpoints(paths, outp, "ties-mean", K=20)
            cm.calculate_metrics(outp)
        meta["order"] = ["vision", "audio"]
        meta["in_configs"] = in_cfg
        meta["merge_metrics"] = open(os.path.join(outp, "merge_metrics.txt")).read()
    _save("g11_metrics", meta=np.array(json.dumps(meta)), **arrays)


def g12():
    """Host-side callers of the path, run from the reference's own code: the length-grouped samplers of the stage-2 trainer
    (modelcompose/train/llava_trainer.py:38-97) for several sizes / seeds, and the LLaVA -> multimodal checkpoint key mapping
    (scripts/convert_llava_to_multimodal/convert_checkpoint.py:47-88) on a tiny two-shard checkpoint."""
    refshim.install()
    import importlib
    sys.modules["peft"].PeftMixedModel = refshim._StubObj("peft.PeftMixedModel")
    import transformers.trainer as tt
    for nm in ("is_sagemaker_mp_enabled", "get_parameter_names", "has_length", "ALL_LAYERNORM_LAYERS", "ShardedDDPOption", "logger"):
        if not hasattr(tt, nm):
            setattr(tt, nm, refshim._StubObj(nm))
    lt = importlib.import_module("modelcompose.train.llava_trainer")
    rng = np.random.default_rng(12)
    meta = {"plain": [], "modality": [], "chunks": []}
    for (n, bs, ws, seed) in ((37, 4, 2, 1), (64, 4, 4, 2), (7, 2, 2, 3), (256, 16, 8, 4), (100, 3, 5, 5)):
        lengths = [int(x) for x in rng.integers(1, 700, n)]
        g = torch.Generator().manual_seed(seed)
        meta["plain"].append({"lengths": lengths, "batch_size": bs, "world_size": ws, "seed": seed,
                              "indices": lt.get_length_grouped_indices(lengths, bs, ws, generator=g)})
    for (n, bs, ws, seed) in ((50, 4, 2, 6), (129, 4, 4, 7), (40, 16, 2, 8)):
        lengths = [int(x) * (1 if rng.random() < 0.6 else -1) for x in rng.integers(1, 700, n)]
        torch.manual_seed(1000 + seed)                      # the per-family shuffles draw from the global RNG (generator=None)
        g = torch.Generator().manual_seed(seed)
        meta["modality"].append({"lengths": lengths, "batch_size": bs, "world_size": ws, "seed": seed, "global_seed": 1000 + seed,
                                 "indices": lt.get_modality_length_grouped_indices(lengths, bs, ws, generator=g)})
    for (n, k) in ((12, 4), (13, 4), (8, 8), (6, 1)):
        lengths = [int(x) for x in rng.integers(1, 50, 40)]
        idx = [int(x) for x in rng.permutation(40)[:n]]
        meta["chunks"].append({"indices": idx, "lengths": lengths, "num_chunks": k, "chunks": lt.split_to_even_chunks(idx, lengths, k)})
    # checkpoint conversion
    sys.path.insert(0, refshim.REF_ROOT + "/scripts/convert_llava_to_multimodal")
    cc = importlib.import_module("convert_checkpoint")
    arrays = {}
    g = torch.Generator().manual_seed(12)
    names1 = ["model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.q_proj.lora_A.default.weight",
              "model.layers.0.self_attn.q_proj.lora_B.default.weight", "model.prefix_tokens", "model.mm_projector.0.weight"]
    names2 = ["model.layers.1.mlp.down_proj.lora_A.default.weight", "model.layers.1.mlp.down_proj.lora_B.default.weight", "model.suffix_tokens",
              "model.mm_projector.0.bias", "model.mm_projector.2.weight", "lm_head.weight", "model.vision_tower.vision_tower.embeddings.cls"]
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "llava")
        os.makedirs(src)
        for fn, names in (("pytorch_model-00001-of-00002.bin", names1), ("pytorch_model-00002-of-00002.bin", names2)):
            w = {k: torch.randn(3, 5, generator=g).to(torch.float16) for k in names}
            torch.save(w, os.path.join(src, fn))
            for k, v in w.items():
                arrays[f"in::{fn}::{k}"] = v.float()
        json.dump({"model_type": "llava", "hidden_size": 8}, open(os.path.join(src, "config.json"), "w"))
        open(os.path.join(src, "tokenizer_config.json"), "w").write("{}")
        open(os.path.join(src, "unrelated.txt"), "w").write("x")
        out = os.path.join(tmp, "out")
        import argparse
        cc.main(argparse.Namespace(llava_checkpoint=src, output_path=out))
        for fn in ("adapter_model.bin", "non_lora_trainables.bin"):
            for k, v in torch.load(os.path.join(out, fn)).items():
                assert v.dtype == torch.float16
                arrays[f"out::{fn}::{k}"] = v.float()
        meta["convert_files"] = sorted(os.listdir(out))
    _save("g12_host", meta=np.array(json.dumps(meta)), **arrays)


def g13():
    """Prompt-side helpers of the path, run from the reference's own modelcompose/mm_utils.py (importable untouched): placeholder ->
    sentinel tokenisation (:43-101), split_string_by_list (:64-79), KeywordsStoppingCriteria (:114-144), expand2square (:14-25),
    get_model_name_from_path (:103-109), with oracle/toy_tokenizer.py standing in for the LLaMA tokenizer."""
    refshim.install()
    import importlib
    from PIL import Image
    from .toy_tokenizer import ToyTokenizer
    mu = importlib.import_module("modelcompose.mm_utils")
    prompts = ["A chat. USER: <image>\nWhat is shown? ASSISTANT:", "<image><audio> both first", "no placeholders at all",
               "USER: <video>\n<point>\n<audio>\n<image>\nDescribe. ASSISTANT:", "ends with <image>", "<image>", "",
               "text <relrep> then <text> then <image> <image>"]
    meta = {"prompts": prompts, "image_token": [], "modal_token": [], "split": [], "stop": [], "names": []}
    for add_bos in (True, False):
        for pr in prompts:
            tok = ToyTokenizer(add_bos)
            meta["image_token"].append({"add_bos": add_bos, "prompt": pr, "ids": mu.tokenizer_image_token(pr, tok)})
            tok = ToyTokenizer(add_bos)
            meta["modal_token"].append({"add_bos": add_bos, "prompt": pr, "ids": mu.tokenizer_modal_token(pr, tok)})
    for pr in prompts:
        meta["split"].append({"prompt": pr, "out": [list(t) for t in mu.split_string_by_list(pr, list(mu.MODAL_TOKEN_MAPPING.keys()))]})
    tok = ToyTokenizer(True)
    prompt_ids = torch.tensor([tok("USER: hello there ASSISTANT:").input_ids])
    for keywords, text in ((["</s>"], "fine thanks </s>"), (["###"], "answer ### more"), (["stop now"], "please stop now ok"), (["never"], "a b c d e")):
        crit = mu.KeywordsStoppingCriteria(keywords, tok, prompt_ids)
        new = tok(text).input_ids[1:]
        verdicts = []
        for n in range(1, len(new) + 1):
            seq = torch.cat([prompt_ids, torch.tensor([new[:n]])], dim=1)
            verdicts.append(bool(crit(seq, None)))
        meta["stop"].append({"keywords": keywords, "text": text, "verdicts": verdicts})
    for pth in ("/ckpts/multimodal-vicuna-7b/", "/a/b/checkpoint-200", "model", "x/y/multimodal-lora/checkpoint-5/"):
        meta["names"].append({"path": pth, "name": mu.get_model_name_from_path(pth)})
    arrays = {}
    rng = np.random.default_rng(13)
    for name, (w, h) in (("wide", (9, 4)), ("tall", (3, 8)), ("square", (5, 5))):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        out = mu.expand2square(Image.fromarray(img), (12, 200, 77))
        arrays[f"img::{name}"] = torch.from_numpy(img.astype(np.int32))
        arrays[f"sq::{name}"] = torch.from_numpy(np.asarray(out).astype(np.int32))
    _save("g13_prompt", meta=np.array(json.dumps(meta)), **arrays)


def g14():
    """Caller-side data formats, run from the reference's own modelcompose/data/utils.py (preprocess: v1 / llama_2 / plain / v0
    templates, with and without placeholders, multi-round, first turn from gpt) and multimodal_dataset.py (the collator: padding,
    truncation, attention mask, per-modality merge order, single-frame video expansion), toy tokenizer / stand-in processors."""
    refshim.install()
    import importlib
    from PIL import Image
    from .toy_tokenizer import ToyTokenizer, FakeProc as _FakeProc
    cl = importlib.import_module("modelcompose.conversation")
    du = importlib.import_module("modelcompose.data.utils")
    convs = {
        "one_round_image": [{"from": "human", "value": "<image>\nWhat is in the picture?"}, {"from": "gpt", "value": "A small cat on a mat."}],
        "two_rounds_audio_video": [{"from": "human", "value": "<video>\n<audio>\nWhat happens here?"}, {"from": "gpt", "value": "A dog barks twice."},
                                   {"from": "human", "value": "Is it loud?"}, {"from": "gpt", "value": "Yes, quite loud."}],
        "gpt_first": [{"from": "gpt", "value": "Hello."}, {"from": "human", "value": "<point>\nDescribe the object."},
                      {"from": "gpt", "value": "A chair with four legs."}],
        "text_only": [{"from": "human", "value": "What is two plus two?"}, {"from": "gpt", "value": "Four."}],
    }
    meta = {"convs": convs, "cases": []}
    arrays = {}
    n = 0
    for tmpl in ("v1", "llava_v1", "llava_llama_2", "plain", "llava_v0"):
        for name, conv in convs.items():
            for has_image in (True, False):
                if tmpl == "plain" and (len(conv) != 2 or not has_image):
                    continue
                if not has_image and name != "text_only":
                    continue
                if has_image and name == "text_only" and tmpl != "v1":
                    continue
                cl.default_conversation = cl.conv_templates[tmpl]
                tok = ToyTokenizer(True, model_max_length=64 if name == "two_rounds_audio_video" and tmpl == "v1" else 2048)
                out = du.preprocess([copy.deepcopy(conv)], tok, has_image=has_image)
                ids, lab = out["input_ids"][0], out["labels"][0]
                arrays[f"ids::{n}"], arrays[f"labels::{n}"] = ids, lab
                meta["cases"].append({"template": tmpl, "conv": name, "has_image": has_image, "max_length": tok.model_max_length, "n": n})
                n += 1
    cl.default_conversation = cl.conv_templates["v1"]
    # collator
    md = importlib.import_module("modelcompose.data.multimodal_dataset")
    tok = ToyTokenizer(True, model_max_length=40)
    rng = np.random.default_rng(14)
    imgs = [rng.integers(0, 256, (4, 6, 3), dtype=np.uint8), rng.integers(0, 256, (5, 5, 3), dtype=np.uint8), rng.integers(0, 256, (7, 3, 3), dtype=np.uint8)]
    vids = [torch.from_numpy(rng.standard_normal((3, 8, 2, 2)).astype(np.float32)), torch.from_numpy(rng.standard_normal((3, 1, 2, 2)).astype(np.float32))]
    pts = [rng.standard_normal((5, 6)).astype(np.float32)]
    samples = [("one_round_image", {"vision": [0]}), ("two_rounds_audio_video", {"video": [0], "audio": ["a.wav"]}),
               ("gpt_first", {"point": [0], "vision": [1, 2], "video": [1]}), ("text_only", {})]
    insts = []
    for cname, mi in samples:
        d = du.preprocess([copy.deepcopy(convs[cname])], tok, has_image=len(mi) != 0)
        inst = {"input_ids": d["input_ids"][0], "labels": d["labels"][0], "modal_inputs": {}}
        for k, v in mi.items():
            inst["modal_inputs"][k] = ([Image.fromarray(imgs[i]) for i in v] if k == "vision" else [vids[i] for i in v] if k == "video"
                                       else [pts[i] for i in v] if k == "point" else list(v))
        insts.append(inst)
    procs = {"vision": _FakeProc("vision"), "audio": _FakeProc("audio"), "point": _FakeProc("point"), "video": None}
    batch = md.DataCollatorForSupervisedDataset(tok, procs, {"vision": {"image_aspect_ratio": "pad"}})(insts)
    for i, a_ in enumerate(imgs):
        arrays[f"col::img::{i}"] = torch.from_numpy(a_.astype(np.int32))
    for i, v in enumerate(vids):
        arrays[f"col::vid::{i}"] = v
    arrays["col::pts::0"] = torch.from_numpy(pts[0])
    arrays["col::input_ids"], arrays["col::labels"], arrays["col::attention_mask"] = batch["input_ids"], batch["labels"], batch["attention_mask"].to(torch.int32)
    arrays["col::vision"], arrays["col::video"], arrays["col::point"] = batch["modal_inputs"]["vision"], batch["modal_inputs"]["video"], batch["modal_inputs"]["point"]
    arrays["col::audio_inputs"] = batch["modal_inputs"]["audio"]["audio_inputs"]
    arrays["col::audio_padding_mask"] = batch["modal_inputs"]["audio"]["audio_padding_mask"].to(torch.int32)
    meta["collate"] = {"samples": [[c, {k: list(v) for k, v in mi.items()}] for c, mi in samples], "max_length": 40,
                       "modal_keys": list(batch["modal_inputs"].keys())}
    _save("g14_data", meta=np.array(json.dumps(meta)), **arrays)


def g15():
    """Full-depth parity fixture.  Case definition: tests/fullwidth_cases.py DEPTH_CASES["fulldepth_iav"] (CPU-generator
    weights and inputs, reproduced bit for bit on the GPU box).  Stored: the new ids, the FULL fp32 logits rows of every step (so any
    top-k / margin can be derived), the fp32 encoder feature blocks' norms and the per-layer hidden-state scale of the last prompt token
    (diagnostics of depth-wise growth).  Follows multimodal_llama.py:488-619 (model forward), eval/model_multimodal_qa_loader.py:94-108
    (greedy generate call)."""
    import time
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    import fullwidth_cases as fc
    from . import pipeline
    torch.set_num_threads(os.cpu_count() or 1)
    for name, fname in (("fulldepth_iav", "g15_fulldepth_iav"), ("depth8_iav", "g15_depth8_iav")):
        t0 = time.time()
        meta, sd, ids, mi = fc.build_case(name)
        fc.sd_to_f32_inplace(sd)
        t1 = time.time()
        with torch.no_grad():
            om = pipeline.OracleModel.from_state_dict(sd, meta)
            new_ids, logits = om.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True)
        t2 = time.time()
        print(f"{name}: weights {t1 - t0:.0f}s oracle {t2 - t1:.0f}s ids {new_ids.tolist()} min margin {fc.margins(logits).min().item():.2e}")
        _save(fname, ids=new_ids, logits=logits.float(), input_ids=ids, margins=fc.margins(logits),
              meta=np.frombuffer(json.dumps({"case": name, "layers": meta["num_hidden_layers"], "seed": fc.DEPTH_CASES[name]["seed"],
                                             "row_seeds": fc.DEPTH_CASES[name]["row_seeds"], "oracle_seconds": round(t2 - t1, 1)}).encode(), dtype=np.uint8))
        del sd, om


def g17(phase=None):
    """Full depth, EIGHT rows, BOTH oracles (VERDICT r3 #2a).  Case: tests/fullwidth_cases.py DEPTH_CASES["fulldepth_iav8"].
      phase A: the fp32 branch-form oracle (oracle/llm.py; multimodal_llama.py:488-619) free-running greedy, two rows at a time (35 GB of
               fp32 weights + the fp32 KV cache of two 2793-token rows); chunk results are kept under /tmp so an interrupted run resumes;
      phase B: the device-rounding restatement (oracle/device_path.py, lazy per-layer composition over the bf16 state dict) TEACHER-FORCED
               on phase A's ids, fed phase A's encoders (fp32 oracle encoders: the backbone is what is compared).
    The two phases run in separate processes (`python -m oracle.gen_golden g17a`, then `g17b`): their weights do not fit together.
    Stored: ids, fp32-oracle logits, device-oracle logits (full fp32 rows), margins, and per-row norms of the oracle's feature blocks."""
    import time
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    import fullwidth_cases as fc
    from . import pipeline
    torch.set_num_threads(os.cpu_count() or 1)
    name = "fulldepth_iav8"
    seeds = fc.DEPTH_CASES[name]["row_seeds"]
    chunks = [seeds[i:i + 2] for i in range(0, len(seeds), 2)]
    tmp = os.environ.get("MC_G17_TMP", "/tmp/mc_g17")
    os.makedirs(tmp, exist_ok=True)
    if phase == "a":
        meta, sd = fc.build_weights(name)
        fc.sd_to_f32_inplace(sd)
        om = pipeline.OracleModel.from_state_dict(sd, meta)
        for ci, rs in enumerate(chunks):
            f = os.path.join(tmp, f"a{ci}.pt")
            if os.path.exists(f):
                continue
            t0 = time.time()
            ids, mi = fc.build_rows(name, rs)
            with torch.no_grad():
                new_ids, logits = om.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True)
            torch.save({"ids": new_ids, "logits": logits.float(), "input_ids": ids, "seconds": time.time() - t0}, f)
            print(f"g17a chunk {ci} rows {rs}: {time.time() - t0:.0f}s ids {new_ids.tolist()}", flush=True)
        return
    if phase == "b":
        meta, sd = fc.build_weights(name)                               # bf16; only what the fp32 encoders touch is widened
        for k in list(sd):
            if sd[k].is_floating_point() and not k.startswith("model.layers.") and k not in ("lm_head.weight",):
                sd[k] = sd[k].float()
        om = pipeline.OracleModel.from_state_dict(sd, meta, emulate="device", device_opts={"lazy": True})
        for ci, rs in enumerate(chunks):
            f = os.path.join(tmp, f"b{ci}.pt")
            if os.path.exists(f):
                continue
            a = torch.load(os.path.join(tmp, f"a{ci}.pt"))
            t0 = time.time()
            ids, mi = fc.build_rows(name, rs)
            assert torch.equal(ids, a["input_ids"])
            with torch.no_grad():
                new_ids, logits = om.generate(ids, fc.to_f32(mi), max_new_tokens=fc.N_NEW, ignore_eos=True, return_logits=True,
                                              forced_ids=a["ids"])
            torch.save({"ids": new_ids, "logits": logits.float(), "seconds": time.time() - t0}, f)
            print(f"g17b chunk {ci} rows {rs}: {time.time() - t0:.0f}s argmax agreement with fp32 "
                  f"{int((new_ids == a['ids']).sum())}/{new_ids.numel()}", flush=True)
        return
    # assemble
    A = [torch.load(os.path.join(tmp, f"a{ci}.pt")) for ci in range(len(chunks))]
    Bd = [torch.load(os.path.join(tmp, f"b{ci}.pt")) for ci in range(len(chunks))]
    ids = torch.cat([a["ids"] for a in A])
    lg = torch.cat([a["logits"] for a in A])
    lgd = torch.cat([b["logits"] for b in Bd])
    _save("g17_fulldepth_iav8", ids=ids, logits=lg, logits_device=lgd, ids_device=torch.cat([b["ids"] for b in Bd]),
          input_ids=torch.cat([a["input_ids"] for a in A]), margins=fc.margins(lg),
          meta=np.frombuffer(json.dumps({"case": name, "layers": 32, "seed": fc.DEPTH_CASES[name]["seed"], "row_seeds": seeds,
                                         "oracle_seconds": [round(a["seconds"], 1) for a in A],
                                         "device_oracle_seconds": [round(b["seconds"], 1) for b in Bd]}).encode(), dtype=np.uint8))


def g16():
    """File-level goldens for the `convert-*` strategies of merge_checkpoints (merge_unimodal_modelcompose.py:42-73): checkpoints trained
    with lora_strategy 'same' (only `.default` adapter keys) are re-labelled 'modal+language', every `.default` tensor of checkpoint i is
    duplicated under its modality's name, and the rest of the strategy string runs on the result: online-merge-reset-*, sum, mean,
    ties-mean, and `drop-mean` (TIES over the shared tensors, then the per-modality copies on top, :62-71)."""
    refshim.install()
    import importlib, io, contextlib
    mm = importlib.import_module("merge_unimodal_modelcompose")
    g = torch.Generator().manual_seed(161)
    shapes = {"model.layers.0.self_attn.q_proj.lora_A.default.weight": (4, 24), "model.layers.0.self_attn.q_proj.lora_B.default.weight": (24, 4),
              "model.layers.1.mlp.down_proj.lora_A.default.weight": (4, 40), "model.layers.1.mlp.down_proj.lora_B.default.weight": (24, 4)}
    order = (("vision", "mm_vision_encoder"), ("audio", "mm_audio_encoder"), ("point", "mm_point_encoder"))
    arrays, meta = {}, {"order": [m for m, _ in order], "cases": {}}
    with tempfile.TemporaryDirectory() as tmp:
        paths, in_cfg = [], {}
        for modal, enc_key in order:
            d = os.path.join(tmp, f"ckpt-{modal}")
            os.makedirs(d)
            w = {k: torch.randn(shp, generator=g) for k, shp in shapes.items()}
            w[f"model.modal_projectors.{modal}.0.weight"] = torch.randn(8, 6, generator=g)
            w[f"prefix_tokens.{modal}"] = torch.randn(1, 2, 8, generator=g)
            torch.save(w, os.path.join(d, "adapter_model.bin"))
            c = {"model_type": "multimodal", enc_key: f"/ckpts/{modal}", "lora_r": 4, "lora_alpha": 8, "lora_strategy": "same",
                 "local_prefix_tokens": 2, "hidden_size": 8}
            json.dump(c, open(os.path.join(d, "config.json"), "w"))
            paths.append(d)
            in_cfg[modal] = c
            for k, v in w.items():
                arrays[f"in::{modal}::{k}"] = v
        meta["in_configs"] = in_cfg
        for tag, strat in (("online", "convert-online-merge-reset-default-vision=0.5,default-audio=0.25,default-point=0.25"),
                           ("online_plain", "convert-online-merge-0.3"), ("sum", "convert-sum"), ("mean", "convert-mean"),
                           ("ties", "convert-ties-mean"), ("drop", "convert-drop-mean"), ("dropsum", "convert-drop-sum")):
            outp = os.path.join(tmp, f"merged-{tag}")
            with contextlib.redirect_stdout(io.StringIO()):
                mm.merge_checkpoints(list(paths), outp, strat, K=20)
            merged = torch.load(os.path.join(outp, "adapter_model.bin"))
            for k, v in merged.items():
                arrays[f"out::{tag}::{k}"] = v
            meta["cases"][tag] = {"strategy": strat, "out_config": json.load(open(os.path.join(outp, "config.json"))),
                                  "merge_info": open(os.path.join(outp, "merge_info.txt")).read().replace(tmp, "<TMP>"),
                                  "keys": list(merged)}
    _save("g16_merge_convert", meta=np.array(json.dumps(meta)), **arrays)


GROUPS = {"g1": g1, "g2": g2, "g3": g3, "g4": g4, "g5_clip": g5_clip, "g5_beats": g5_beats, "g5_qformer": g5_qformer, "g5_video": g5_video,
          "g5_point": g5_point, "g5_imagebind": g5_imagebind, "g6": g6, "g7": g7, "g8": g8, "g9": g9, "g10": g10, "g11": g11, "g12": g12, "g13": g13, "g14": g14, "g16": g16, "g18": g18}
SLOW_GROUPS = {"g15": g15, "g17a": lambda: g17("a"), "g17b": lambda: g17("b"), "g17": g17}          # by name only


def main(argv):
    if argv and all(a in SLOW_GROUPS for a in argv):
        for a in argv:
            print(f"== {a}")
            SLOW_GROUPS[a]()
        return
    refshim.install()
    todo = argv or list(GROUPS)
    for g in todo:
        print(f"== {g}")
        GROUPS[g]()


if __name__ == "__main__":
    main(sys.argv[1:])
