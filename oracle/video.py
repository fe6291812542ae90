"""Oracle (TEST INFRASTRUCTURE): LanguageBind video preprocessing after decoding.

Reference: modelcompose/model/multimodal_encoder/languagebind/video/processing_video.py:24-68 (transform), :107-129 (frame
sampling with np.linspace(0, n-1, num_frames, dtype=int)).  The transform is a Compose of third-party ops that are not installed
here (torchvision / pytorchvideo): x / 255.0, NormalizeVideo(mean, std), ShortSideScale(224) — pytorchvideo's
torch.nn.functional.interpolate(size=..., mode='bilinear', align_corners=False) with the long side floor(long / short * 224) —
CenterCropVideo(224) — torchvision's crop at int(round((h - 224) / 2.0)) — and RandomHorizontalFlipVideo(p=0.5), a RANDOM flip that
the reference leaves on at inference; here `flip` is an explicit argument.  The arithmetic is restated on top of the same torch
primitives those libraries call (torch IS installed), so the pin is torch's own interpolate."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def sample_frame_ids(duration: int, num_frames: int = 8) -> np.ndarray:
    return np.linspace(0, duration - 1, num_frames, dtype=int)                      # :115, :123


def video_transform(frames_u8: torch.Tensor, size: int = 224, flip: bool = False) -> torch.Tensor:
    """frames_u8 [T, H, W, 3] uint8 (decord / cv2 layout) -> [3, T, size, size] float32."""
    x = frames_u8.permute(3, 0, 1, 2)                                                # (T, H, W, C) -> (C, T, H, W), :117
    x = x / 255.0
    mean = torch.tensor(OPENAI_DATASET_MEAN, dtype=torch.float32).view(3, 1, 1, 1)
    std = torch.tensor(OPENAI_DATASET_STD, dtype=torch.float32).view(3, 1, 1, 1)
    x = (x - mean) / std                                                             # NormalizeVideo
    c, t, h, w = x.shape
    if w < h:                                                                        # ShortSideScale
        new_h, new_w = int(math.floor((float(h) / w) * size)), size
    else:
        new_h, new_w = size, int(math.floor((float(w) / h) * size))
    x = F.interpolate(x, size=(new_h, new_w), mode="bilinear", align_corners=False)
    i, j = int(round((new_h - size) / 2.0)), int(round((new_w - size) / 2.0))        # CenterCropVideo
    x = x[..., i:i + size, j:j + size]
    if flip:
        x = x.flip(-1)
    return x.contiguous()
