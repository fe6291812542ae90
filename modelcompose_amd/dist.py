"""Data-parallel eval sharding.  The reference runs N independent processes over contiguous chunks of the question
list and concatenates their files (eval/model_multimodal_qa_loader.py:25-46, scripts/model_composition/test/MCUB-4.sh:42-70).
Same partition rule here; the only exchange is the final all-gather of generated ids (RCCL over xGMI)."""
from __future__ import annotations

import math
from typing import List, Sequence

import torch


def split_list(lst: Sequence, n: int) -> List[Sequence]:
    """model_multimodal_qa_loader.py:25-28."""
    chunk = math.ceil(len(lst) / n)
    return [lst[i:i + chunk] for i in range(0, len(lst), chunk)]


def get_chunk(lst: Sequence, n: int, k: int):
    """model_multimodal_qa_loader.py:31-33."""
    return split_list(lst, n)[k]


def gather_ids(ids: torch.Tensor, world_size: int, force: bool = False) -> torch.Tensor:
    """All ranks' generated ids [B, T] -> [world*B, T], rank-major = the order of the reference's `cat` of chunk files.
    force: run the collective even in a world of one (exercises the RCCL path on a single GPU)."""
    if world_size == 1 and not force:
        return ids
    import torch.distributed as dist
    out = torch.empty((world_size * ids.shape[0],) + tuple(ids.shape[1:]), dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(out, ids.contiguous())
    return out
