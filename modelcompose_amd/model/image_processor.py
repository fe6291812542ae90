"""CLIP image processor on the HIP path — host-side mirror of what the vision branch does before the tower:
`expand2square` (modelcompose/mm_utils.py:14-26) and transformers' CLIPImageProcessor.preprocess as loaded by
modelcompose/model/multimodal_encoder/clip_encoder.py:53 (resize shortest edge with PIL bicubic, centre crop, rescale 1/255,
normalise).  The pixel work runs in csrc/preprocess.hip on the uint8 image already in HBM; only Pillow's small coefficient tables
(a few KB, double precision -> 22-bit fixed point) are built on the host, exactly as Pillow's precompute_coeffs does."""
from __future__ import annotations

import ctypes as C
import math
from functools import lru_cache
from typing import Dict, Sequence, Union

import numpy as np
import torch

from .. import _lib

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
_PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@lru_cache(maxsize=256)
def _coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter: bounds [out, 2] (first index, count), kk [out, ksize]."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(n)]
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds[xx] = (xmin, n)
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << _PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PRECISION_BITS))
    return bounds, kk, ksize


class HipCLIPImageProcessor:
    """Field-compatible with the parts of transformers.CLIPImageProcessor the reference touches: `.image_mean`, `.crop_size`,
    `.preprocess(image, return_tensors='pt')['pixel_values']`, `__call__(images, return_tensors='pt')`."""

    def __init__(self, size: int = 336, crop_size: int = None, image_mean=OPENAI_CLIP_MEAN, image_std=OPENAI_CLIP_STD, device="cuda",
                 out_dtype=torch.float32):
        self.size = {"shortest_edge": size}
        c = size if crop_size is None else crop_size
        self.crop_size = {"height": c, "width": c}
        self.image_mean, self.image_std = list(image_mean), list(image_std)
        self.device, self.out_dtype = torch.device(device), out_dtype
        self._tables: Dict = {}

    @classmethod
    def from_pretrained(cls, path, **kw):
        import json
        import os
        d = json.load(open(os.path.join(path, "preprocessor_config.json")))
        size = d.get("size", 224)
        size = size.get("shortest_edge", size.get("height")) if isinstance(size, dict) else size
        crop = d.get("crop_size", size)
        crop = crop.get("height") if isinstance(crop, dict) else crop
        return cls(size, crop, d.get("image_mean", OPENAI_CLIP_MEAN), d.get("image_std", OPENAI_CLIP_STD), **kw)

    def _dev_tables(self, in_size, out_size):
        key = (in_size, out_size)
        if key not in self._tables:
            b, k, ks = _coeffs(in_size, out_size)
            self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).contiguous().to(self.device), ks)
        return self._tables[key]

    def _one(self, image, pad_color=None, out_u8: bool = False):
        """image: PIL.Image (RGB) or uint8 array / tensor [H, W, 3].  pad_color: expand2square background (None = no padding)."""
        if not torch.cuda.is_available():
            raise RuntimeError("HipCLIPImageProcessor runs on the HIP device; no CPU fallback")
        if hasattr(image, "convert"):
            image = np.array(image.convert("RGB"))                 # a writable copy (np.asarray of a PIL image is read-only)
        img = torch.as_tensor(image)
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("image must be uint8 [H, W, 3]")
        img = img.to(self.device).contiguous()
        h, w = int(img.shape[0]), int(img.shape[1])
        if pad_color is not None and h != w:
            s = max(h, w)
            ch, cw = s, s
            off_y, off_x = ((w - h) // 2, 0) if w > h else (0, (h - w) // 2)          # mm_utils.py:18-26
        else:
            ch, cw, off_y, off_x = h, w, 0, 0
        size = self.size["shortest_edge"]
        short, long_ = (cw, ch) if cw <= ch else (ch, cw)
        new_long = int(size * long_ / short)                                          # transformers get_resize_output_image_size
        rh, rw = (new_long, size) if cw <= ch else (size, new_long)
        crop_h, crop_w = self.crop_size["height"], self.crop_size["width"]
        if crop_h > rh or crop_w > rw:
            raise NotImplementedError("centre crop larger than the resized image (zero padding branch of transformers.center_crop)")
        top, left = (rh - crop_h) // 2, (rw - crop_w) // 2
        bh = kh = bv = kv = None
        ksh = ksv = 0
        if rw != cw:
            bh, kh, ksh = self._dev_tables(cw, rw)
        if rh != ch:
            bv, kv, ksv = self._dev_tables(ch, rh)
        tmp = torch.empty(ch * rw * 3, dtype=torch.uint8, device=self.device)
        out = torch.empty(3, crop_h, crop_w, dtype=self.out_dtype, device=self.device)
        u8 = torch.empty(crop_h, crop_w, 3, dtype=torch.uint8, device=self.device) if out_u8 else None
        bg = (C.c_int32 * 3)(*([int(c) for c in pad_color] if pad_color is not None else [0, 0, 0]))
        mean = (C.c_float * 3)(*self.image_mean)
        std = (C.c_float * 3)(*self.image_std)
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(_lib.lib().mc_image_preprocess_u8(
            img.data_ptr(), h, w, ch, cw, off_y, off_x, bg, p(bh), p(kh), ksh, p(bv), p(kv), ksv, rh, rw, top, left, crop_h, crop_w, mean, std,
            tmp.data_ptr(), out.data_ptr() if self.out_dtype == _lib.storage_dtype() else None,
            out.data_ptr() if self.out_dtype == torch.float32 else None, p(u8), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "mc_image_preprocess_u8")
        return (out, u8) if out_u8 else out

    def preprocess(self, images, return_tensors="pt", pad_to_square: bool = False, **kw):
        single = hasattr(images, "convert") or (hasattr(images, "ndim") and images.ndim == 3) or (torch.is_tensor(images) and images.dim() == 3)
        lst = [images] if single else list(images)
        pad = tuple(int(x * 255) for x in self.image_mean) if pad_to_square else None      # mm_utils.py:33
        outs = [self._one(im, pad) for im in lst]
        return {"pixel_values": torch.stack(outs, 0)}

    __call__ = preprocess
