"""LanguageBindVideoProcessor on the HIP path — host-side mirror of
modelcompose/model/multimodal_encoder/languagebind/video/processing_video.py:24-68, :107-129, :134-170 for frames that are already
decoded (decord / cv2 / pytorchvideo decoding stays with the data pipeline).  One kernel does /255, normalisation, the bilinear
short-side resize, the centre crop and the (explicit) horizontal flip."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from .. import _lib

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def sample_frame_ids(duration: int, num_frames: int = 8) -> np.ndarray:
    """Frame indices the decord / opencv back ends read (:115, :123)."""
    return np.linspace(0, duration - 1, num_frames, dtype=int)


class HipLanguageBindVideoProcessor:
    def __init__(self, num_frames: int = 8, size: int = 224, device="cuda", out_dtype=None):
        self.num_frames, self.size, self.device = num_frames, size, torch.device(device)
        self.out_dtype = _lib.storage_dtype() if out_dtype is None else out_dtype

    def transform(self, frames_u8: torch.Tensor, flip: bool = False) -> torch.Tensor:
        """frames_u8 [T, H, W, 3] uint8 -> [3, T, size, size].  `flip`: the reference's transform ends with
        RandomHorizontalFlipVideo(p=0.5) even at inference (:42, :54, :66); the caller draws that coin."""
        if not torch.cuda.is_available():
            raise RuntimeError("HipLanguageBindVideoProcessor runs on the HIP device; no CPU fallback")
        f = torch.as_tensor(frames_u8)
        if f.dtype != torch.uint8 or f.dim() != 4 or f.shape[3] != 3:
            raise ValueError("frames must be uint8 [T, H, W, 3]")
        f = f.to(self.device).contiguous()
        T, H, W = int(f.shape[0]), int(f.shape[1]), int(f.shape[2])
        s = self.size
        if W < H:                                                    # pytorchvideo short_side_scale
            rh, rw = int(math.floor((float(H) / W) * s)), s
        else:
            rh, rw = s, int(math.floor((float(W) / H) * s))
        top, left = int(round((rh - s) / 2.0)), int(round((rw - s) / 2.0))          # torchvision center_crop
        out = torch.empty(3, T, s, s, dtype=self.out_dtype, device=self.device)
        mean, std = (C.c_float * 3)(*OPENAI_DATASET_MEAN), (C.c_float * 3)(*OPENAI_DATASET_STD)
        _lib.check(_lib.lib().mc_video_preprocess_u8(f.data_ptr(), T, H, W, rh, rw, top, left, s, 1 if flip else 0, mean, std,
                                                     out.data_ptr() if self.out_dtype == _lib.storage_dtype() else None,
                                                     out.data_ptr() if self.out_dtype == torch.float32 else None,
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mc_video_preprocess_u8")
        return out

    def __call__(self, images=None, text=None, return_tensors=None, flip: bool = False, **kw):
        if text is not None:
            raise NotImplementedError("the text branch of LanguageBindVideoProcessor (:150-152) is never used by the composed model")
        if images is None:
            raise ValueError("You have to specify either text or images. Both cannot be none.")            # :147
        lst = images if isinstance(images, (list, tuple)) else [images]
        if any(isinstance(v, str) for v in lst):
            raise NotImplementedError("video decoding (decord / cv2, :107-129) is left to the data pipeline: pass the sampled frames "
                                      "[num_frames, H, W, 3] uint8 (indices: sample_frame_ids)")
        return {"pixel_values": torch.stack([self.transform(v, flip) for v in lst])}

    def preprocess(self, images, return_tensors=None):
        return self(images=images, return_tensors=return_tensors)
