"""ImageBind audio branch on the HIP path — the audio encoder selected when the encoder path contains 'VideoLLaMA'
(modelcompose/model/multimodal_encoder/builder.py:91-95 -> imagebind_huge, imagebind/imagebind_model.py:548-567).

Host-side mirror of ImageBindModel.forward / get_audio_feature (imagebind_model.py:455-527) restricted to the audio modality
(the only one the reference ever calls): AudioPreprocessor (multimodal_preprocessors.py:120-160, 205-316), the SimpleTransformer
trunk with nn.MultiheadAttention(add_bias_kv=True) (imagebind/transformer.py:94-97, 105-173, 176-280), head and postprocessor
(imagebind_model.py:402-406, 436-439).  Every tensor op is a kernel of libmc_hip.so."""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch

from ..checkpoint_io import load_tensors

from .. import _lib, ops

BF16 = _lib.storage_dtype()      # the library's 16-bit storage element: bf16, or fp16 with MC_STORAGE_DTYPE=fp16 (_lib.set_storage_dtype)


class ImageBindAudioConfig:
    """imagebind_huge audio defaults (imagebind_model.py:40-76, 548-561)."""

    def __init__(self, audio_kernel_size=16, audio_stride=10, audio_embed_dim=768, audio_num_blocks=12, audio_num_heads=12,
                 audio_num_mel_bins=128, audio_target_len=204, out_embed_dim=1024, **kw):
        self.audio_kernel_size, self.audio_stride, self.audio_embed_dim = audio_kernel_size, audio_stride, audio_embed_dim
        self.audio_num_blocks, self.audio_num_heads = audio_num_blocks, audio_num_heads
        self.audio_num_mel_bins, self.audio_target_len, self.out_embed_dim = audio_num_mel_bins, audio_target_len, out_embed_dim


class HipImageBindAudioEncoder:
    def __init__(self, audio_encoder: Optional[str], args=None, delay_load=False, config: Optional[dict] = None, device="cuda"):
        self.audio_encoder_name, self.device, self.dtype = audio_encoder, torch.device(device), BF16
        self.cfg = ImageBindAudioConfig(**(config or {}))
        self.is_loaded = False
        self.modal_processor = None                     # the reference installs nn.Identity() (multimodal_encoder/builder.py:93-94)
        ck = None if audio_encoder is None else os.path.join(str(audio_encoder), "imagebind_huge.pth")
        self.ckpt_path = ck if ck is not None and os.path.exists(ck) else None
        if not delay_load and self.ckpt_path:
            self.load_model()

    config = property(lambda self: self.cfg)
    hidden_size = property(lambda self: self.cfg.out_embed_dim)         # "audio_hidden_size=1024" (imagebind_model.py:567)

    @property
    def dummy_inputs(self):
        c = self.cfg
        return torch.zeros(1, 3, 1, c.audio_num_mel_bins, c.audio_target_len, device=self.device, dtype=self.dtype)

    def load_model(self):
        if not self.is_loaded:
            if not self.ckpt_path:
                raise FileNotFoundError("Can't load ImageBindModel, since ckpt_path is invalid.")      # imagebind_model.py:545
            self.load_state_dict(load_tensors(self.ckpt_path))

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        c, dev = self.cfg, self.device
        E = c.audio_embed_dim
        t = lambda k: sd[k].to(dev, BF16).contiguous()
        pp, tr = "modality_preprocessors.audio.", "modality_trunks.audio."
        self.patch_w = ops.pack_weight(sd[pp + "rgbt_stem.proj.weight"].to(dev).reshape(E, -1))          # conv, no bias
        self.stem_ln = (t(pp + "rgbt_stem.norm_layer.weight"), t(pp + "rgbt_stem.norm_layer.bias"))
        self.cls = sd[pp + "cls_token"].to(dev, BF16).reshape(-1).contiguous()
        self.pos = sd[pp + "pos_embedding_helper.pos_embed"].to(dev, BF16).reshape(-1, E).contiguous()
        self.blocks = []
        for i in range(c.audio_num_blocks):
            p = f"{tr}blocks.{i}."
            self.blocks.append(dict(
                n1=(t(p + "norm_1.weight"), t(p + "norm_1.bias")), n2=(t(p + "norm_2.weight"), t(p + "norm_2.bias")),
                qkv=ops.pack_weight(sd[p + "attn.in_proj_weight"].to(dev), sd[p + "attn.in_proj_bias"].to(dev)),
                bias_kv=torch.cat([sd[p + "attn.bias_k"].reshape(1, E), sd[p + "attn.bias_v"].reshape(1, E)], 1).to(dev, BF16).contiguous(),
                out=ops.pack_weight(sd[p + "attn.out_proj.weight"].to(dev), sd[p + "attn.out_proj.bias"].to(dev)),
                fc1=ops.pack_weight(sd[p + "mlp.fc1.weight"].to(dev), sd[p + "mlp.fc1.bias"].to(dev)),
                fc2=ops.pack_weight(sd[p + "mlp.fc2.weight"].to(dev), sd[p + "mlp.fc2.bias"].to(dev))))
        self.head_ln = (t("modality_heads.audio.0.weight"), t("modality_heads.audio.0.bias"))
        self.head = ops.pack_weight(sd["modality_heads.audio.2.weight"].to(dev))
        # Normalize(dim=-1) followed by the fixed logit scale: y * scale / ||y||  =  rmsnorm(y) * scale / sqrt(out_dim)
        scale = min(float(sd["modality_postprocessors.audio.1.log_logit_scale"].float().exp()), 100.0)
        O = self.head.N
        self.post_w = torch.full((O,), scale / math.sqrt(O), dtype=torch.float32).to(dev, BF16)
        self.is_loaded = True

    def __call__(self, inputs):
        return self.forward(inputs)

    def forward(self, inputs: torch.Tensor, return_cls: bool = False) -> torch.Tensor:
        """(B, S, 1, mel, frames) -> (B, S, out_embed_dim)   (forward = get_audio_feature(...)[1], imagebind_model.py:455-460)."""
        c, dev = self.cfg, self.device
        E, H = c.audio_embed_dim, c.audio_num_heads
        d = E // H
        x = inputs.to(dev, BF16)
        if x.dim() != 5:
            raise ValueError("ImageBind audio input must be (B, clips, 1, mel_bins, frames)")
        B, S = x.shape[:2]
        n = B * S
        x = x.reshape(n, *x.shape[2:]).contiguous()                                  # clips folded into the batch (:496-503)
        k, s = c.audio_kernel_size, c.audio_stride
        cols, oh, ow = ops.im2col(x, k, k, s, s)
        T = oh * ow
        if T + 1 != self.pos.shape[0]:
            raise ValueError(f"{T} patches but the position table has {self.pos.shape[0] - 1}: interpolation is not implemented")
        patches = ops.linear(cols, self.patch_w)
        patches = ops.layernorm(patches, self.stem_ln[0], self.stem_ln[1], 1e-5)
        Tt = T + 1
        h = ops.vit_assemble(patches, self.cls, self.pos, n, T, E).view(n * Tt, E)
        # K/V buffer with the learned bias_k / bias_v token appended after the input projection (add_bias_kv)
        kv = torch.empty(n * (Tt + 1), 2 * E, dtype=BF16, device=dev)
        r = torch.arange(n * Tt, device=dev, dtype=torch.int32)
        dst_rows = (r + r // Tt).contiguous()                                         # sample b, token t -> row b*(Tt+1) + t
        bias_rows = (torch.arange(n, device=dev, dtype=torch.int32) * (Tt + 1) + Tt).contiguous()
        zeros = torch.zeros(n, device=dev, dtype=torch.int32)
        a = torch.empty(n * Tt, E, dtype=BF16, device=dev)
        for L in self.blocks:
            y = ops.layernorm(h, L["n1"][0], L["n1"][1], 1e-6)
            qkv = ops.linear(y, L["qkv"])
            ops.copy_rows(qkv[:, E:], kv, n * Tt, None, dst_rows)
            ops.copy_rows(L["bias_kv"], kv, n, zeros, bias_rows)
            ks = ((Tt + 1) * 2 * E, 2 * E, d)
            ops.attn_prefill(qkv, kv, kv[:, E:], a, n, H, H, Tt, Tt + 1, d, (Tt * 3 * E, 3 * E, d), ks, ks, E, False, 0, scale=d ** -0.5)
            h = ops.linear(a, L["out"], residual=h)
            y = ops.layernorm(h, L["n2"][0], L["n2"][1], 1e-6)
            h = ops.linear(ops.linear(y, L["fc1"], act="gelu"), L["fc2"], residual=h)
        cls_rows = torch.empty(n, E, dtype=BF16, device=dev)
        ops.copy_rows(h, cls_rows, n, (torch.arange(n, device=dev, dtype=torch.int32) * Tt).contiguous(), None)   # SelectElement(0)
        cls = ops.layernorm(cls_rows, self.head_ln[0], self.head_ln[1], 1e-6)
        y = ops.linear(_padk(cls, self.head.Kp), self.head)
        y = ops.rmsnorm(y, self.post_w, 1e-24)
        y = y.view(B, S, -1)
        return (cls.view(B, S, -1), y) if return_cls else y


def _padk(x: torch.Tensor, Kp: int) -> torch.Tensor:
    return x if x.shape[1] == Kp else torch.nn.functional.pad(x, (0, Kp - x.shape[1]))
