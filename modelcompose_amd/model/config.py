"""MultimodalConfig and the LocalLoRA adapter plan.

Host-side mirror of modelcompose/model/language_model/multimodal_llama.py:33-61 (config defaults persisted in
config.json), multimodal_encoder/builder.py:119-129 (infer_modals: adapter order default, audio, vision, video,
point) and LocalLoraLinear.__init__/extract_params (:84-118: adapter set, 'default-{m}' adapters, scaling)."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Tuple


class MultimodalConfig:
    model_type = "multimodal"
    # class-level defaults of the reference (multimodal_llama.py:33-61)
    _defaults = dict(
        vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
        num_key_value_heads=None, max_position_embeddings=4096, rms_norm_eps=1e-5, rope_theta=10000.0, hidden_act="silu",
        pad_token_id=0, bos_token_id=1, eos_token_id=2, pretraining_tp=1, rope_scaling=None,
        lora_strategy=None, lora_name="default", lora_r=128, lora_alpha=256, lora_dropout=0.05,
        local_prefix_tokens=0, local_suffix_tokens=0,
        local_vision_prefix_tokens=None, local_vision_suffix_tokens=None, local_audio_prefix_tokens=None,
        local_audio_suffix_tokens=None, local_video_prefix_tokens=None, local_video_suffix_tokens=None,
        local_point_prefix_tokens=None, local_point_suffix_tokens=None,
        layer_local_tokens=False, seperate_layernorm=False, merge_default_weights=None, reset_scaling_weights=None,
        mm_vision_encoder=None, mm_vision_tower=None, mm_audio_encoder=None, mm_video_encoder=None, mm_point_encoder=None,
        mm_projector_type="linear", mm_vision_select_layer=-2, mm_vision_select_feature="patch",
    )

    def __init__(self, **kw):
        for k, v in self._defaults.items():
            setattr(self, k, v)
        for k, v in kw.items():
            setattr(self, k, v)
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        if self.rope_scaling is not None:
            raise ValueError(f"Unknown RoPE scaling type {self.rope_scaling}: only rope_scaling=None is supported")  # :191-205
        if self.pretraining_tp not in (None, 1):
            raise ValueError("pretraining_tp > 1 (weight-slicing emulation, :222-235) is not supported; Vicuna uses 1")

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @classmethod
    def from_pretrained(cls, path: str) -> "MultimodalConfig":
        with open(os.path.join(path, "config.json")) as f:
            return cls(**json.load(f))

    def to_dict(self) -> dict:
        return {k: v for k, v in self.__dict__.items() if not k.startswith("_")}

    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2)


def infer_modals(cfg) -> List[str]:
    """multimodal_encoder/builder.py:119-129."""
    modals = ["default"]
    if getattr(cfg, "mm_audio_encoder", None) is not None:
        modals.append("audio")
    if getattr(cfg, "mm_vision_encoder", None) is not None or getattr(cfg, "mm_vision_tower", None) is not None:
        modals.append("vision")
    if getattr(cfg, "mm_video_encoder", None):
        modals.append("video")
    if getattr(cfg, "mm_point_encoder", None):
        modals.append("point")
    return modals


def extract_params(s: str) -> Dict[str, float]:
    """multimodal_llama.py:109-118; malformed strings raise ValueError like the reference's unpack/float()."""
    out = {}
    for pair in s.split(","):
        key, value = pair.split("=")
        out[key.strip()] = float(value)
    return out


def adapter_plan(cfg) -> Tuple[List[str], Dict[str, float], Optional[List[str]], Optional[str]]:
    """(adapter names, scaling, default_adapter_names, merge mode) — LocalLoraLinear.__init__ :84-107."""
    names = infer_modals(cfg)
    scaling = {n: cfg.lora_alpha / cfg.lora_r for n in names}
    merge = cfg.merge_default_weights
    default_names = None
    if cfg.reset_scaling_weights is not None:
        reset = extract_params(cfg.reset_scaling_weights)
        if any("default-" in k for k in reset):
            merge = "linear-"
            default_names = [f"default-{n}" for n in names[1:]]
            for dn in default_names:
                names.append(dn)
                scaling[dn] = cfg.lora_alpha / cfg.lora_r
        for k in reset:
            if k in scaling:
                scaling[k] = scaling[k] * reset[k]
    return names, scaling, default_names, merge


def composition_terms(cfg, adapter: str, has_lora) -> List[Tuple[str, float]]:
    """Which (lora adapter key, scale) terms make the dense weight of routed adapter `adapter`
    (LocalLoraLinear.forward :120-160).  has_lora(key) says whether lora_A/B for that key exist."""
    names, scaling, default_names, merge = adapter_plan(cfg)
    if adapter not in names:
        return []                                                        # :127-129 falls back to the base output
    if adapter == "default" and merge is not None:                       # :130-149
        if not (merge in ("sum", "mean") or merge.startswith("linear-")):
            raise NotImplementedError(f"online merging strategy '{merge}' is not implemented.")
        div = len(default_names) if merge == "mean" else 1
        # adapters whose weights were never loaded keep B = 0 after reset_lora_parameters (builder.py:150-153)
        return [(dn, scaling[dn] / div) for dn in default_names if has_lora(dn)]
    if not has_lora(adapter):
        return []
    return [(adapter, scaling[adapter])]                                  # :150-157
