"""Synthetic (seeded, random-init) checkpoints in the reference's state_dict key grammar, generated directly in
HBM — there is no network for real weights: weights N(0, 0.02²), LoRA A kaiming-uniform(a=√5),
LoRA B N(0, 0.01²) (non-zero so composition matters), prefix/suffix N(0, 0.02²), norm gains 1."""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch


def vicuna7b_meta(modals: Sequence[str] = ("vision",), reset: Optional[str] = None, layers: int = 32, prefix: int = 5) -> dict:
    meta = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=layers, num_attention_heads=32,
                num_key_value_heads=32, max_position_embeddings=4096, rms_norm_eps=1e-5, lora_r=128, lora_alpha=256,
                lora_strategy="modal+language", reset_scaling_weights=reset, local_prefix_tokens=prefix, local_suffix_tokens=prefix,
                pad_token_id=0, eos_token_id=2, mm_projector_type="mlp2x_gelu", mm_vision_select_layer=-2,
                mm_vision_select_feature="patch")
    for m in modals:
        meta[f"mm_{m}_encoder"] = f"synthetic/{m}"
    if "vision" in modals:
        meta["clip"] = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                            patch_size=14, layer_norm_eps=1e-5, hidden_act="quick_gelu")
    order = ["default"] + [m for m in ("audio", "vision", "video", "point") if m in modals]
    meta["modal_names"] = order
    return meta


def synthetic_state_dict(meta: dict, device="cuda", seed: int = 1234, dtype=torch.bfloat16, lora_adapters: Optional[Sequence[str]] = None
                         ) -> Dict[str, torch.Tensor]:
    g = torch.Generator(device=device).manual_seed(seed)
    Hd, I, V, Lyr = meta["hidden_size"], meta["intermediate_size"], meta["vocab_size"], meta["num_hidden_layers"]
    H, Hkv = meta["num_attention_heads"], meta["num_key_value_heads"]
    D = Hd // H
    r = meta["lora_r"]
    names = meta["modal_names"]
    if lora_adapters is None:
        lora_adapters = list(names)
        if meta.get("reset_scaling_weights") and "default-" in meta["reset_scaling_weights"]:
            lora_adapters += [f"default-{m}" for m in names[1:]]

    def nrm(*shape, std=0.02):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

    def kaiming(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)           # kaiming_uniform_(a=sqrt(5)) on [out, in]
        return ((torch.rand(out_f, in_f, generator=g, device=device, dtype=torch.float32) * 2 - 1) * bound).to(dtype)

    sd = {"model.embed_tokens.weight": nrm(V, Hd), "lm_head.weight": nrm(V, Hd),
          "model.norm.weight": torch.ones(Hd, device=device, dtype=dtype)}
    shapes = {"self_attn.q_proj": (H * D, Hd), "self_attn.k_proj": (Hkv * D, Hd), "self_attn.v_proj": (Hkv * D, Hd),
              "self_attn.o_proj": (Hd, H * D), "mlp.gate_proj": (I, Hd), "mlp.up_proj": (I, Hd), "mlp.down_proj": (Hd, I)}
    for l in range(Lyr):
        p = f"model.layers.{l}"
        sd[f"{p}.input_layernorm.weight"] = torch.ones(Hd, device=device, dtype=dtype)
        sd[f"{p}.post_attention_layernorm.weight"] = torch.ones(Hd, device=device, dtype=dtype)
        for lin, (n, k) in shapes.items():
            sd[f"{p}.{lin}.weight"] = nrm(n, k)
            for ad in lora_adapters:
                sd[f"{p}.{lin}.lora_A.{ad}.weight"] = kaiming(r, k)
                sd[f"{p}.{lin}.lora_B.{ad}.weight"] = nrm(n, r, std=0.01)
    npre = meta.get("local_prefix_tokens", 0)
    if npre:
        for m in names:
            sd[f"prefix_tokens.{m}"] = nrm(1, npre, Hd)
            sd[f"suffix_tokens.{m}"] = nrm(1, meta.get("local_suffix_tokens", npre), Hd)
    if "clip" in meta:
        c = meta["clip"]
        Dm, Im, T = c["hidden_size"], c["intermediate_size"], (c["image_size"] // c["patch_size"]) ** 2 + 1
        pre = "model.modal_encoders.vision.vision_tower.vision_model."
        sd[pre + "embeddings.class_embedding"] = nrm(Dm)
        sd[pre + "embeddings.patch_embedding.weight"] = nrm(Dm, 3, c["patch_size"], c["patch_size"])
        sd[pre + "embeddings.position_embedding.weight"] = nrm(T, Dm)
        for nm in ("pre_layrnorm", "post_layernorm"):
            sd[pre + nm + ".weight"] = torch.ones(Dm, device=device, dtype=dtype)
            sd[pre + nm + ".bias"] = torch.zeros(Dm, device=device, dtype=dtype)
        for i in range(c["num_hidden_layers"]):
            q = f"{pre}encoder.layers.{i}."
            for nm in ("layer_norm1", "layer_norm2"):
                sd[q + nm + ".weight"] = torch.ones(Dm, device=device, dtype=dtype)
                sd[q + nm + ".bias"] = torch.zeros(Dm, device=device, dtype=dtype)
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sd[q + f"self_attn.{nm}.weight"] = nrm(Dm, Dm)
                sd[q + f"self_attn.{nm}.bias"] = nrm(Dm)
            sd[q + "mlp.fc1.weight"] = nrm(Im, Dm); sd[q + "mlp.fc1.bias"] = nrm(Im)
            sd[q + "mlp.fc2.weight"] = nrm(Dm, Im); sd[q + "mlp.fc2.bias"] = nrm(Dm)
        sd["model.modal_projectors.vision.0.weight"] = nrm(Hd, Dm); sd["model.modal_projectors.vision.0.bias"] = nrm(Hd)
        sd["model.modal_projectors.vision.2.weight"] = nrm(Hd, Hd); sd["model.modal_projectors.vision.2.bias"] = nrm(Hd)
    return sd


def synthetic_prompt(B: int, modal_sentinels: Sequence[int], vocab: int = 32000, n_before: int = 35, n_after: int = 60, seed: int = 0):
    """[1] + n_before ids + (sentinel, 13)* + n_after ids, ids U{3..vocab-1}, equal length across the batch."""
    g = torch.Generator().manual_seed(seed)
    parts = [torch.ones(B, 1, dtype=torch.long), torch.randint(3, vocab, (B, n_before), generator=g)]
    for s in modal_sentinels:
        parts += [torch.full((B, 1), s, dtype=torch.long), torch.full((B, 1), 13, dtype=torch.long)]
    parts.append(torch.randint(3, vocab, (B, n_after), generator=g))
    return torch.cat(parts, 1)
