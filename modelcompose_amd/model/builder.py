"""load_pretrained_model — host-side mirror of modelcompose/model/builder.py:27-231 (multimodal branch :138-185):
same signature, same dispatch rule ('multimodal' in model_name.lower()), same files read
(config.json, base shards, adapter_model.bin | mm_projector.bin, non_lora_trainables.bin), same return tuple
(tokenizer, model, modal_processors, context_len).  Weights end up composed + packed in HBM (bf16)."""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Optional

import torch

from ..checkpoint_io import load_tensors

from .clip import ClipVisionConfig, HipClipVisionTower
from .config import MultimodalConfig, infer_modals
from .multimodal_llama import MultimodalLlamaForCausalLM
from .projector import build_projector, projector_type_for


def _load_file(path: str) -> Dict[str, torch.Tensor]:
    return load_tensors(path)            # native reader: torch zip checkpoints and safetensors alike (checkpoint_io.py)


def load_base_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """HF checkpoint directory -> flat state dict (single file, sharded .bin with index, or safetensors)."""
    sd: Dict[str, torch.Tensor] = {}
    for pat in ("model.safetensors", "pytorch_model.bin"):
        p = os.path.join(path, pat)
        if os.path.exists(p):
            return _load_file(p)
    for idx in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        p = os.path.join(path, idx)
        if os.path.exists(p):
            files = sorted(set(json.load(open(p))["weight_map"].values()))
            for f in files:
                sd.update(_load_file(os.path.join(path, f)))
            return sd
    files = sorted(glob.glob(os.path.join(path, "pytorch_model-*.bin")) + glob.glob(os.path.join(path, "model-*.safetensors")))
    if not files:
        raise FileNotFoundError(f"no model weights found under {path}")
    for f in files:
        sd.update(_load_file(f))
    return sd


def build_modal_modules(model: MultimodalLlamaForCausalLM, clip_config: Optional[ClipVisionConfig] = None, delay_load: bool = True,
                        encoder_hidden: Optional[Dict[str, int]] = None, encoder_configs: Optional[Dict[str, dict]] = None):
    """multimodal_encoder/builder.py:86-117 (build_modal_encoders) + multimodal_projector/builder.py:246-260."""
    cfg = model.config
    dev = model.device
    encoder_hidden = encoder_hidden or {}
    encoder_configs = encoder_configs or {}
    for modal in [m for m in infer_modals(cfg) if m != "default"]:
        if modal == "vision":
            path = getattr(cfg, "mm_vision_tower", None) or cfg.mm_vision_encoder
            enc = HipClipVisionTower(path if (path and os.path.isdir(path)) else None, cfg, delay_load=delay_load, config=clip_config,
                                     device=dev)
            enc.vision_tower_name = path
            hidden = enc.config.hidden_size if enc.config is not None else getattr(cfg, "mm_hidden_size", None)
        else:
            from . import encoders_extra
            enc, hidden = encoders_extra.build(modal, cfg, dev, delay_load, config=encoder_configs.get(modal))
        model.model.modal_encoders[modal] = enc
        hidden = encoder_hidden.get(modal, hidden)
        key = "mm_hidden_size" if modal == "vision" else f"mm_{modal}_hidden_size"
        hidden = getattr(cfg, key, None) or hidden
        model.model.modal_projectors[modal] = _build_modal_projector(cfg, modal, hidden, dev, encoder_configs.get("qformer"))
    return model


def _build_modal_projector(cfg, modal, hidden, dev, qformer_config=None):
    if modal == "audio" and "VideoLLaMA" in str(getattr(cfg, "mm_audio_encoder", "")):
        from . import encoders_extra
        return encoders_extra.build_audio_qformer(cfg, dev, num_positions=8, config=qformer_config)   # builder.py:249-251
    ptype = projector_type_for(cfg, modal)
    if ptype.startswith("qformer"):
        from . import encoders_extra
        return encoders_extra.build_qformer_projector(cfg, ptype, hidden, dev, config=qformer_config)
    return build_projector(ptype, hidden, cfg.hidden_size, dev)


def get_model_name_from_path(model_path: str) -> str:
    """mm_utils.py:103-109."""
    model_path = model_path.strip("/")
    parts = model_path.split("/")
    if parts[-1].startswith("checkpoint-"):
        return parts[-2] + "_" + parts[-1]
    return parts[-1]


def load_pretrained_model(model_path, model_base, model_name, load_8bit=False, load_4bit=False, device_map="auto", device="cuda", *,
                          torch_dtype=None):
    """torch_dtype (extension; the reference hard-codes torch.float16, builder.py:41): None = the process's storage dtype (bf16 unless
    MC_STORAGE_DTYPE=fp16); torch.float16 / "fp16" selects the IEEE-half instantiation of the library for the whole process."""
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes quantised loading (builder.py:30-39) is out of scope of the HIP path")
    if torch_dtype is not None:
        from .. import _lib
        _lib.set_storage_dtype(torch_dtype)
    if "multimodal" not in model_name.lower():
        raise ValueError(f"model name '{model_name}' does not contain 'multimodal': only the composed-model branch "
                         f"(builder.py:138-185) is implemented")
    cfg = MultimodalConfig.from_pretrained(model_path)                                   # :142
    model = MultimodalLlamaForCausalLM(cfg, device=device)
    build_modal_modules(model, delay_load=True)
    base = model_base if model_base is not None else model_path
    model.load_state_dict(load_base_state_dict(base))                                    # from_pretrained(model_base) :148
    if model_base is not None:
        adapter_path = os.path.join(model_path, "adapter_model.bin")                     # :157-162
        if not os.path.exists(adapter_path):
            adapter_path = os.path.join(model_path, "mm_projector.bin")
        model.load_state_dict(load_tensors(adapter_path))
        nl = os.path.join(model_path, "non_lora_trainables.bin")                         # :164-168
        if os.path.exists(nl):
            model.load_state_dict(load_tensors(nl))
    for modal, enc in model.model.modal_encoders.items():                                # :180-183
        enc.load_model()
    model.finalize()
    tokenizer = None
    try:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(base, use_fast=False)
    except Exception:
        tokenizer = None          # tokenizer files are optional for tensor-level use
    modal_processors = model.get_modal_processors()                                      # :223-224
    context_len = getattr(cfg, "max_sequence_length", 2048)                              # :226-229
    return tokenizer, model, modal_processors, context_len


def build_from_state_dict(meta: dict, sd: Dict[str, torch.Tensor], device="cuda", dtype=None) -> MultimodalLlamaForCausalLM:
    """Construct a model from an in-memory reference-grammar state dict (tests, smoke, synthetic benchmarks).
    dtype: None = the process's storage dtype (bf16 unless MC_STORAGE_DTYPE=fp16 / set_storage_dtype); "fp16" / torch.float16 selects the
    IEEE-half instantiation of the library - the reference's own inference dtype (model/builder.py:41, :162, :185) - for the WHOLE process:
    one storage dtype per process, so models built before with the other one must not be used any more."""
    if dtype is not None:
        from .. import _lib
        _lib.set_storage_dtype(dtype)
    known = set(MultimodalConfig._defaults)
    cfg = MultimodalConfig(**{k: v for k, v in meta.items() if k in known or k.startswith("mm_") or k.startswith("local_")})
    model = MultimodalLlamaForCausalLM(cfg, device=device)
    clip_cfg = ClipVisionConfig(**meta["clip"]) if "clip" in meta else None
    enc_cfgs = {m: meta[k] for m, k in (("audio", "beats"), ("audio", "imagebind"), ("video", "video"), ("point", "point"),
                                        ("qformer", "qformer")) if k in meta}
    build_modal_modules(model, clip_config=clip_cfg, delay_load=True, encoder_configs=enc_cfgs)
    if "fps_start" in meta and "point" in model.model.modal_encoders:
        model.model.modal_encoders["point"].fps_start = torch.as_tensor(meta["fps_start"])
    model.load_state_dict(sd)
    model.finalize()
    return model
