"""Host-side mirror of the helpers of modelcompose/mm_utils.py that sit on the path (image batch preparation, model name)."""
from __future__ import annotations

import torch


def process_images(images, image_processor, model_cfg):
    """mm_utils.py:29-44: image_aspect_ratio == 'pad' -> expand2square with the processor's mean colour, then preprocess; else the
    processor on the whole list.  With the HIP processor the padding is part of the same kernels (a virtual canvas)."""
    aspect = getattr(model_cfg, "image_aspect_ratio", None)
    if aspect == "pad":
        new_images = [image_processor.preprocess(im, return_tensors="pt", pad_to_square=True)["pixel_values"][0] for im in images]
        if all(x.shape == new_images[0].shape for x in new_images):
            new_images = torch.stack(new_images, dim=0)
        return new_images
    return image_processor(images, return_tensors="pt")["pixel_values"]


def get_model_name_from_path(model_path: str) -> str:
    """mm_utils.py:103-109."""
    model_path = model_path.strip("/")
    parts = model_path.split("/")
    if parts[-1].startswith("checkpoint-"):
        return parts[-2] + "_" + parts[-1]
    return parts[-1]
