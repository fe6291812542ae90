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


def gather_ids(ids: torch.Tensor, world_size: int, force: bool = False, pad_value: int = 0, equal_shapes: bool = False,
               return_rows: bool = False):
    """All ranks' generated ids -> one tensor, rank-major = the order of the reference's `cat` of chunk files
    (eval/model_multimodal_qa_loader.py:25-33, scripts/model_composition/test/MCUB-4.sh:60-70).

    Ranks may hold DIFFERENT shapes: the last `ceil(n/N)` chunk is shorter (or empty) and a batch stops at its own longest row (EOS), so
    [B_r, T_r] differs per rank.  One small all-gather exchanges the shapes; if they differ every rank pads to [max B, max T] with
    `pad_value` (HF pads finished rows with pad_token_id the same way), one all-gather moves the ids, and the padding ROWS are dropped so
    that the result is [sum_r B_r, max T] in rank order.  `equal_shapes=True` (fixed-shape benchmark loops) skips the shape exchange.
    force: run the collectives even in a world of one (exercises the RCCL path on a single GPU).
    return_rows: also return the per-rank row counts (list of int)."""
    if world_size == 1 and not force:
        return (ids, [int(ids.shape[0])]) if return_rows else ids
    import torch.distributed as dist
    assert ids.dim() == 2, "gather_ids takes [rows, tokens]"
    ids = ids.contiguous()
    if ids.is_cuda and dist.get_backend() == "gloo":
        # gloo moves device tensors for broadcast / all-reduce only: gather on the host (functional multi-process tests on a box with fewer GPUs
        # than ranks; the measured path is RCCL)
        res = gather_ids(ids.cpu(), world_size, force=force, pad_value=pad_value, equal_shapes=equal_shapes, return_rows=return_rows)
        return (res[0].to(ids.device), res[1]) if return_rows else res.to(ids.device)
    if equal_shapes:
        out = torch.empty((world_size * ids.shape[0], ids.shape[1]), dtype=ids.dtype, device=ids.device)
        dist.all_gather_into_tensor(out, ids)
        return (out, [int(ids.shape[0])] * world_size) if return_rows else out
    mine = torch.tensor([ids.shape[0], ids.shape[1]], dtype=torch.int64, device=ids.device)
    shapes = torch.empty(world_size * 2, dtype=torch.int64, device=ids.device)
    dist.all_gather_into_tensor(shapes, mine)
    shapes = shapes.view(world_size, 2).cpu()
    rows = [int(r) for r in shapes[:, 0]]
    Bm, Tm = int(shapes[:, 0].max()), int(shapes[:, 1].max())
    if Bm == 0 or Tm == 0:
        out = torch.empty((sum(rows), Tm), dtype=ids.dtype, device=ids.device)
        return (out, rows) if return_rows else out
    if tuple(ids.shape) != (Bm, Tm):
        padded = torch.full((Bm, Tm), pad_value, dtype=ids.dtype, device=ids.device)
        padded[:ids.shape[0], :ids.shape[1]] = ids
    else:
        padded = ids
    out = torch.empty((world_size * Bm, Tm), dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(out, padded)
    if any(r != Bm for r in rows):
        out = torch.cat([out[k * Bm:k * Bm + r] for k, r in enumerate(rows)], 0)
    return (out, rows) if return_rows else out


def gather_rows(t: torch.Tensor, world_size: int, force: bool = False, equal_shapes: bool = False) -> torch.Tensor:
    """All ranks' rows of a [rows, ...] tensor (any dtype; trailing dimensions equal on every rank) -> [sum_r rows_r, ...] in rank order.
    Ranks may hold different row counts (the last chunk of `ceil(n/N)` is shorter or empty): the counts are exchanged first and the
    padding rows dropped, as gather_ids does for the id matrix."""
    if world_size == 1 and not force:
        return t
    import torch.distributed as dist
    t = t.contiguous()
    if t.is_cuda and dist.get_backend() == "gloo":
        return gather_rows(t.cpu(), world_size, force=force, equal_shapes=equal_shapes).to(t.device)
    tail = tuple(t.shape[1:])
    if equal_shapes:
        out = torch.empty((world_size * t.shape[0],) + tail, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out
    mine = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = torch.empty(world_size, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(counts, mine)
    rows = [int(r) for r in counts.cpu()]
    Bm = max(rows)
    if Bm == 0:
        return torch.empty((0,) + tail, dtype=t.dtype, device=t.device)
    padded = t
    if t.shape[0] != Bm:
        padded = torch.zeros((Bm,) + tail, dtype=t.dtype, device=t.device)
        padded[:t.shape[0]] = t
    out = torch.empty((world_size * Bm,) + tail, dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, padded)
    if any(r != Bm for r in rows):
        out = torch.cat([out[k * Bm:k * Bm + r] for k, r in enumerate(rows)], 0)
    return out


def gather_logits(logits: torch.Tensor, world_size: int, force: bool = False, equal_shapes: bool = False) -> torch.Tensor:
    """BASELINE.json's north_star words the eval aggregation as an all-gather of LOGITS: every rank's step logits [B_r, T, V] fp32 ->
    [sum_r B_r, T, V] in rank order (the reference itself aggregates text files, scripts/model_composition/test/MCUB-4.sh:60-70; gather_ids
    moves what those files hold).  One RCCL all_gather_into_tensor over xGMI, T V 4 bytes per row."""
    assert logits.dim() == 3, "gather_logits takes [rows, steps, vocab]"
    return gather_rows(logits, world_size, force=force, equal_shapes=equal_shapes)
