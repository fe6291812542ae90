"""Gradient buckets of the data-parallel training step (host logic, importable without a GPU).

The flat fp32 gradient buffer is laid out in backward order (last decoder layer first, projector / prefix / suffix tokens last), so
the bucket of layers [l, l+k) is complete as soon as layer l's backward has run and its all-reduce overlaps the remaining layers.
The reference leaves this to DeepSpeed ZeRO-2/3 (scripts/model_composition/train/run_finetune_*_damc.sh:27-30)."""
from __future__ import annotations

from typing import Dict, List, Tuple


def bucket_ranges(layer_end: Dict[int, int], n_layers: int, bucket_layers: int, n_params: int) -> List[Tuple[int, int, int]]:
    """[(ready_after_layer, lo, hi)] in completion order; the last entry (ready_after_layer = -1) holds everything behind layer 0."""
    out = []
    for l in reversed(range(n_layers)):
        if l % bucket_layers == 0:
            lo = 0 if l + bucket_layers >= n_layers else layer_end[l + bucket_layers]
            out.append((l, lo, layer_end[l]))
    out.append((-1, layer_end[0], n_params))
    return [b for b in out if b[2] > b[1]]


def allreduce_buckets(flat_grad, ranges, group=None):
    """Launch one asynchronous SUM all-reduce per bucket (torch.distributed: RCCL on the GPU box, gloo in the CPU tests)."""
    import torch.distributed as dist
    return [dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True) for (_, lo, hi) in ranges]
