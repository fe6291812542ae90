"""Oracle (TEST INFRASTRUCTURE): image preprocessing of the vision branch.

Reference path: modelcompose/mm_utils.py:14-40 (`expand2square` with the processor's mean colour when image_aspect_ratio ==
'pad', then `image_processor.preprocess`) where the processor is transformers' CLIPImageProcessor loaded by
modelcompose/model/multimodal_encoder/clip_encoder.py:53 — resize (shortest edge, PIL bicubic), centre crop, rescale 1/255,
normalise.  PIL and transformers are third-party code outside /root/reference; both are installed here, so this restatement is
pinned against PIL.Image.resize / CLIPImageProcessor themselves in tests/test_oracle_golden.py.

PIL's 8-bit resampling (Pillow src/libImaging/Resample.c) is restated exactly: per output index a window
[center - support, center + support] (support = 2 * max(1, in/out) for bicubic), double-precision weights normalised to sum 1 and
converted to 22-bit fixed point, integer accumulation with a rounding bias and a clamp to 0..255; horizontal pass first, then
vertical, both rounding to uint8."""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int):
    """precompute_coeffs + normalize_coeffs_8bpc: (bounds [out, 2] = (first input index, count), coeffs [out, ksize] int32)."""
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
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        bounds[xx] = (xmin, xmax)
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
    return bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    bounds, kk = resample_coeffs(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img [H, W, C] uint8 -> [out_h, out_w, C] uint8, PIL Image.resize((out_w, out_h), resample=BICUBIC) semantics (a pass whose
    size does not change is skipped, as ImagingResample does)."""
    x = img
    if out_w != img.shape[1]:
        x = _resample_axis(x, out_w, 1)
    if out_h != img.shape[0]:
        x = _resample_axis(x, out_h, 0)
    return x


def expand2square(img: np.ndarray, background) -> np.ndarray:
    """mm_utils.py:14-26 on an [H, W, C] uint8 array."""
    h, w = img.shape[:2]
    if w == h:
        return img
    s = max(w, h)
    out = np.empty((s, s, img.shape[2]), dtype=np.uint8)
    out[:] = np.asarray(background, dtype=np.uint8)
    if w > h:
        out[(w - h) // 2:(w - h) // 2 + h] = img
    else:
        out[:, (h - w) // 2:(h - w) // 2 + w] = img
    return out


def clip_preprocess(img: np.ndarray, size: int = 336, pad: bool = True, mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD) -> np.ndarray:
    """[H, W, 3] uint8 -> [3, size, size] float32: (expand2square with int(mean*255) when pad) -> resize shortest edge to `size`
    (the long edge scales with int(size * long / short), transformers get_resize_output_image_size) -> centre crop -> /255 -> normalise."""
    if pad:
        img = expand2square(img, tuple(int(m * 255) for m in mean))
    h, w = img.shape[:2]
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    oh, ow = (new_long, new_short) if w <= h else (new_short, new_long)
    x = resize_bicubic_u8(img, oh, ow)
    top, left = (oh - size) // 2, (ow - size) // 2
    x = x[top:top + size, left:left + size]
    f = (x.astype(np.float64) * (1 / 255)).astype(np.float32)
    f = (f - np.asarray(mean, dtype=np.float32)) / np.asarray(std, dtype=np.float32)
    return np.ascontiguousarray(f.transpose(2, 0, 1))
