"""Checkpoint file access for the loaders: the files the reference reads with torch.load / safetensors
(builder.py:138-185: sharded `pytorch_model-0000x-of-0000y.bin`, `adapter_model.bin`, `non_lora_trainables.bin`, encoder checkpoints).

torch's zip checkpoints are memory-mapped (`mmap=True`): tensors alias the page cache instead of being unpickled into anonymous
memory, so the 13.5 GB base never exists twice on the host and each tensor's bytes are touched once, by its host-to-device copy.
`weights_only=True` keeps unpickling to tensors and plain containers (checkpoints are untrusted input).  Legacy (pre-zip) files cannot
be mapped and fall back to a plain restricted load."""
from __future__ import annotations

from typing import Any

import torch


def load_tensors(path: str) -> Any:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    try:
        return torch.load(path, map_location="cpu", mmap=True, weights_only=True)
    except (RuntimeError, ValueError) as e:
        if "mmap" not in str(e).lower():
            raise
        return torch.load(path, map_location="cpu", weights_only=True)
