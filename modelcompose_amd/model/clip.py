"""CLIP-ViT vision tower on the HIP path.

Host-side mirror of modelcompose/model/multimodal_encoder/clip_encoder.py:9-119 (CLIPVisionTower: HF
CLIPVisionModel, hidden_states[select_layer], drop the class token for 'patch').  The transformer itself
(transformers==4.31 CLIPVisionTransformer: conv patch-embed without bias, class token, learned positions,
pre_layrnorm, 24 x [LN, MHA(16x64, biases, q*d^-0.5), +res, LN, fc1, QuickGELU, fc2, +res]) runs as:
im2col + MFMA GEMM, LayerNorm kernel, fused-QKV GEMM, flash attention, GEMMs with bias/activation/residual
epilogues.  Layers after the selected hidden state are never computed (the reference computes and discards
the last layer, clip_encoder.py:59-60)."""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

from ..checkpoint_io import load_tensors

from .. import _lib, ops

BF16 = _lib.storage_dtype()      # the library's 16-bit storage element: bf16, or fp16 with MC_STORAGE_DTYPE=fp16 (_lib.set_storage_dtype)


class ClipVisionConfig:
    def __init__(self, hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                 image_size=336, patch_size=14, num_channels=3, layer_norm_eps=1e-5, hidden_act="quick_gelu", **kw):
        self.hidden_size, self.intermediate_size = hidden_size, intermediate_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.image_size, self.patch_size, self.num_channels = image_size, patch_size, num_channels
        self.layer_norm_eps, self.hidden_act = layer_norm_eps, hidden_act

    @classmethod
    def from_pretrained(cls, path):
        d = json.load(open(os.path.join(path, "config.json")))
        d = d.get("vision_config", d)
        return cls(**d)


def _strip(sd: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


class HipClipVisionTower:
    """Drop-in for CLIPVisionTower: tower(images) -> (B, n_patches, hidden) features."""

    def __init__(self, vision_tower: Optional[str], args=None, delay_load: bool = False, config: Optional[ClipVisionConfig] = None,
                 device="cuda"):
        self.is_loaded = False
        self.vision_tower_name = vision_tower
        self.select_layer = getattr(args, "mm_vision_select_layer", -2) if args is not None else -2
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch") if args is not None else "patch"
        self.device = torch.device(device)
        self.dtype = BF16
        self.config = config
        self.image_processor = None
        if config is None and vision_tower is not None and os.path.isdir(vision_tower):
            self.config = ClipVisionConfig.from_pretrained(vision_tower)
        if not delay_load and vision_tower is not None:
            self.load_model()

    # -- reference surface -------------------------------------------------------------------------
    @property
    def hidden_size(self):
        return self.config.hidden_size

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2

    @property
    def dummy_inputs(self):
        c = self.config
        return torch.zeros(1, c.num_channels, c.image_size, c.image_size, device=self.device, dtype=self.dtype)

    @property
    def modal_processor(self):
        if self.image_processor is None and self.config is not None:
            from .image_processor import HipCLIPImageProcessor
            self.image_processor = HipCLIPImageProcessor(self.config.image_size, device=self.device)
        return self.image_processor

    def load_model(self):
        if self.is_loaded:
            return
        path = self.vision_tower_name
        sd = None
        for fn in ("model.safetensors", "pytorch_model.bin"):
            p = os.path.join(path, fn)
            if os.path.exists(p):
                sd = load_tensors(p)               # native reader: safetensors and torch zip alike
                break
        if sd is None:
            raise FileNotFoundError(f"no CLIP weights (model.safetensors / pytorch_model.bin) under {path}")
        self.load_state_dict(sd)
        # clip_encoder.py:53 loads transformers' CLIPImageProcessor (PIL on the CPU); here the same steps run on the device
        from .image_processor import HipCLIPImageProcessor
        try:
            self.image_processor = HipCLIPImageProcessor.from_pretrained(path, device=self.device)
        except Exception:
            self.image_processor = HipCLIPImageProcessor(self.config.image_size, device=self.device)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Accepts HF CLIPVisionModel keys ('vision_model.…', as in 4.31 checkpoints) or the un-prefixed 5.x grammar."""
        if any(k.startswith("vision_model.") for k in sd):
            sd = _strip(sd, "vision_model.")
        c, dev = self.config, self.device
        D = c.hidden_size
        t = lambda k: sd[k].to(dev, BF16).contiguous()
        self.patch_w = ops.pack_weight(sd["embeddings.patch_embedding.weight"].to(dev).reshape(D, -1))
        self.cls = t("embeddings.class_embedding")
        self.pos = t("embeddings.position_embedding.weight")
        self.pre_ln = (t("pre_layrnorm.weight"), t("pre_layrnorm.bias"))
        self.layers = []
        for i in range(c.num_hidden_layers):
            p = f"encoder.layers.{i}."
            if p + "self_attn.q_proj.weight" not in sd:
                break
            # fused QKV; the q scale d^-0.5 (CLIPAttention) is applied as the softmax scale of the attention kernel
            wqkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0).to(dev)
            bqkv = torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0).to(dev)
            self.layers.append(dict(
                ln1=(t(p + "layer_norm1.weight"), t(p + "layer_norm1.bias")),
                ln2=(t(p + "layer_norm2.weight"), t(p + "layer_norm2.bias")),
                qkv=ops.pack_weight(wqkv, bqkv),
                out=ops.pack_weight(sd[p + "self_attn.out_proj.weight"].to(dev), sd[p + "self_attn.out_proj.bias"].to(dev)),
                fc1=ops.pack_weight(sd[p + "mlp.fc1.weight"].to(dev), sd[p + "mlp.fc1.bias"].to(dev)),
                fc2=ops.pack_weight(sd[p + "mlp.fc2.weight"].to(dev), sd[p + "mlp.fc2.bias"].to(dev)),
            ))
        self.is_loaded = True

    # -- forward -----------------------------------------------------------------------------------
    def hidden_state(self, images: torch.Tensor, index: int) -> torch.Tensor:
        """hidden_states[index] of the HF tuple (0 = embeddings after pre-LN, k = output of layer k)."""
        c = self.config
        D, H = c.hidden_size, c.num_attention_heads
        d = D // H
        x = images.to(self.device, BF16)
        B = x.shape[0]
        cols, oh, ow = ops.im2col(x, c.patch_size, c.patch_size, c.patch_size, c.patch_size)
        T = oh * ow
        patches = ops.linear(cols, self.patch_w)
        h = ops.vit_assemble(patches, self.cls, self.pos, B, T, D).view(B * (T + 1), D)
        h = ops.layernorm(h, self.pre_ln[0], self.pre_ln[1], c.layer_norm_eps)
        Tt = T + 1
        st = (Tt * 3 * D, 3 * D, d)
        for i in range(index):
            L = self.layers[i]
            n = ops.layernorm(h, L["ln1"][0], L["ln1"][1], c.layer_norm_eps)
            qkv = ops.linear(n, L["qkv"])
            a = torch.empty(B * Tt, D, dtype=BF16, device=self.device)
            ops.attn_prefill(qkv, qkv[:, D:], qkv[:, 2 * D:], a, B, H, H, Tt, Tt, d, st, st, st, D, False, 0, scale=d ** -0.5)
            h = ops.linear(a, L["out"], residual=h)
            n = ops.layernorm(h, L["ln2"][0], L["ln2"][1], c.layer_norm_eps)
            f = ops.linear(n, L["fc1"], act=c.hidden_act)
            h = ops.linear(f, L["fc2"], residual=h)
        return h.view(B, Tt, D)

    def __call__(self, images):
        return self.forward(images)

    def forward(self, images):
        if type(images) is list:                                     # clip_encoder.py:71-76
            return [self.forward(im.unsqueeze(0)) for im in images]
        n_hs = self.config.num_hidden_layers + 1
        idx = self.select_layer if self.select_layer >= 0 else n_hs + self.select_layer
        f = self.hidden_state(images, idx)
        if self.select_feature == "patch":                            # clip_encoder.py:59-67
            f = f[:, 1:]
        elif self.select_feature != "cls_patch":
            raise ValueError(f"Unexpected select feature: {self.select_feature}")
        return f
