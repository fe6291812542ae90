"""Modal projectors on the HIP path — mirror of modelcompose/model/multimodal_projector/builder.py:202-260
(build_vision_projector: 'linear' / 'mlpNx_gelu' / 'identity'; per-modality wrappers that read
mm_{modal}_projector_type / mm_{modal}_hidden_size)."""
from __future__ import annotations

import re
from typing import Dict

import torch

from .. import _lib, ops

BF16 = _lib.storage_dtype()      # the library's 16-bit storage element: bf16, or fp16 with MC_STORAGE_DTYPE=fp16 (_lib.set_storage_dtype)


class HipMlpProjector:
    """Linear -> (GELU -> Linear)*  with exact-erf GELU fused in the GEMM epilogue."""

    def __init__(self, in_features: int, hidden: int, depth: int, device="cuda"):
        self.in_features, self.hidden, self.depth, self.device = in_features, hidden, depth, torch.device(device)
        self.weights = []

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        self.weights = []
        keys = ["weight"] if self.depth == 0 else [f"{2 * i}.weight" for i in range(self.depth)]
        for k in keys:
            w = sd[prefix + k].to(self.device)
            b = sd[prefix + k.replace("weight", "bias")].to(self.device)
            self.weights.append(ops.pack_weight(w, b))

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        h = x.reshape(-1, shp[-1]).to(BF16)
        if not h.is_contiguous():
            h = h.contiguous()
        Kp = self.weights[0].Kp
        if h.shape[1] != Kp:
            h = torch.nn.functional.pad(h, (0, Kp - h.shape[1]))
        n = len(self.weights)
        for i, w in enumerate(self.weights):
            h = ops.linear(h, w, act="gelu" if i < n - 1 else None)
        return h.view(*shp[:-1], h.shape[-1])


class IdentityMap:
    def load_state_dict(self, sd, prefix=""):
        pass

    def __call__(self, x):
        return x


def build_projector(projector_type: str, mm_hidden_size: int, hidden_size: int, device="cuda"):
    """builder.py:202-226; unknown types raise ValueError like the reference (:226)."""
    if projector_type == "linear":
        return HipMlpProjector(mm_hidden_size, hidden_size, 0, device)
    m = re.match(r"^mlp(\d+)x_gelu$", projector_type)
    if m:
        return HipMlpProjector(mm_hidden_size, hidden_size, int(m.group(1)), device)
    if projector_type == "identity":
        return IdentityMap()
    raise ValueError(f"Unknown projector type: {projector_type}")


def projector_type_for(cfg, modal: str) -> str:
    """builder.py:228-244: vision uses mm_projector_type, others mm_{modal}_projector_type (default 'linear')."""
    if modal == "vision":
        return getattr(cfg, "mm_projector_type", "linear") or "linear"
    return getattr(cfg, f"mm_{modal}_projector_type", "linear") or "linear"
