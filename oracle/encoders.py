"""Oracle (TEST INFRASTRUCTURE): modality encoders + projectors, torch CPU fp32.

CLIP-ViT follows modelcompose/model/multimodal_encoder/clip_encoder.py:59-81 and
the transformers CLIPVisionModel it wraps (third-party, transformers==4.31;
restated from its published behaviour: conv patch-embed without bias, class
token, learned positions, pre-LN, LN->MHA->res->LN->fc1->QuickGELU->fc2->res,
hidden_states[k] = input of layer k+1).

Projectors follow modelcompose/model/multimodal_projector/builder.py:202-226.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F


@dataclass
class ClipVisionConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 336
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"


def _act(x, name):
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return F.gelu(x)
    raise ValueError(name)


def mha(x, sd, pre, n_heads, names=("q_proj", "k_proj", "v_proj", "out_proj"), kv=None, mask=None):
    """CLIPAttention (transformers 4.31): q scaled by head_dim**-0.5, biases, softmax fp32."""
    B, L, Dm = x.shape
    kv = x if kv is None else kv
    S = kv.shape[1]
    d = Dm // n_heads
    q = F.linear(x, sd[f"{pre}.{names[0]}.weight"], sd.get(f"{pre}.{names[0]}.bias")) * (d ** -0.5)
    k = F.linear(kv, sd[f"{pre}.{names[1]}.weight"], sd.get(f"{pre}.{names[1]}.bias"))
    v = F.linear(kv, sd[f"{pre}.{names[2]}.weight"], sd.get(f"{pre}.{names[2]}.bias"))
    q = q.view(B, L, n_heads, d).transpose(1, 2)
    k = k.view(B, S, n_heads, d).transpose(1, 2)
    v = v.view(B, S, n_heads, d).transpose(1, 2)
    w = q @ k.transpose(-1, -2)
    if mask is not None:
        w = w + mask
    w = F.softmax(w, dim=-1)
    o = (w @ v).transpose(1, 2).reshape(B, L, Dm)
    return F.linear(o, sd[f"{pre}.{names[3]}.weight"], sd.get(f"{pre}.{names[3]}.bias"))


def clip_vision_hidden_states(pixels: torch.Tensor, sd: Dict[str, torch.Tensor], cfg: ClipVisionConfig,
                              prefix: str = "vision_model", n_layers: Optional[int] = None):
    """Returns the HF hidden_states tuple (embeddings after pre-LN, then each layer output)."""
    B = pixels.shape[0]
    w = sd[f"{prefix}.embeddings.patch_embedding.weight"]
    x = F.conv2d(pixels, w, bias=None, stride=cfg.patch_size)            # (B, D, g, g)
    x = x.flatten(2).transpose(1, 2)
    cls = sd[f"{prefix}.embeddings.class_embedding"].expand(B, 1, -1)
    x = torch.cat([cls, x], dim=1) + sd[f"{prefix}.embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (cfg.hidden_size,), sd[f"{prefix}.pre_layrnorm.weight"], sd[f"{prefix}.pre_layrnorm.bias"],
                     cfg.layer_norm_eps)
    hs = [x]
    L = cfg.num_hidden_layers if n_layers is None else n_layers
    for i in range(L):
        p = f"{prefix}.encoder.layers.{i}"
        r = x
        x = F.layer_norm(x, (cfg.hidden_size,), sd[f"{p}.layer_norm1.weight"], sd[f"{p}.layer_norm1.bias"], cfg.layer_norm_eps)
        x = r + mha(x, sd, f"{p}.self_attn", cfg.num_attention_heads)
        r = x
        x = F.layer_norm(x, (cfg.hidden_size,), sd[f"{p}.layer_norm2.weight"], sd[f"{p}.layer_norm2.bias"], cfg.layer_norm_eps)
        x = F.linear(x, sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"])
        x = _act(x, cfg.hidden_act)
        x = r + F.linear(x, sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"])
        hs.append(x)
    return hs


def clip_vision_tower(pixels, sd, cfg: ClipVisionConfig, select_layer: int = -2, select_feature: str = "patch",
                      prefix: str = "vision_model"):
    """CLIPVisionTower.forward + feature_select (clip_encoder.py:59-81).

    hidden_states has num_hidden_layers+1 entries; index -2 = output of layer
    L-1 (the last layer's output is computed by the reference and discarded, so
    only select_layer-dependent layers are evaluated here)."""
    n_hs = cfg.num_hidden_layers + 1
    idx = select_layer if select_layer >= 0 else n_hs + select_layer
    hs = clip_vision_hidden_states(pixels, sd, cfg, prefix, n_layers=idx)
    f = hs[idx]
    if select_feature == "patch":
        f = f[:, 1:]
    elif select_feature != "cls_patch":
        raise ValueError(f"Unexpected select feature: {select_feature}")
    return f


def mlp_projector(x, sd, prefix, depth=2):
    """build_vision_projector 'mlpNx_gelu' (multimodal_projector/builder.py:208-215): Linear, (GELU, Linear)*."""
    x = F.linear(x, sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"])
    for i in range(1, depth):
        x = F.gelu(x)
        x = F.linear(x, sd[f"{prefix}.{2 * i}.weight"], sd[f"{prefix}.{2 * i}.bias"])
    return x


def projector(x, sd, prefix, projector_type: str):
    """build_vision_projector dispatch (multimodal_projector/builder.py:202-226)."""
    if projector_type == "linear":
        return F.linear(x, sd[f"{prefix}.weight"], sd[f"{prefix}.bias"])
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        return mlp_projector(x, sd, prefix, int(m.group(1)))
    if projector_type == "identity":
        return x
    raise ValueError(f"Unknown projector type: {projector_type}")
