"""Length-grouped batch sampler of the stage-2 finetune (host logic; integer / index work, bit-exact with the reference for the same
torch generator state).  Mirrors modelcompose/train/llava_trainer.py:38-57 (split_to_even_chunks), :60-86
(get_modality_length_grouped_indices), :89-97 (get_length_grouped_indices) and :133-165 (LengthGroupedSampler)."""
from __future__ import annotations

from typing import List, Optional

import torch
from torch.utils.data import Sampler


def split_to_even_chunks(indices: List[int], lengths, num_chunks: int) -> List[List[int]]:
    """Greedy balance: every index goes to the chunk with the smallest total length so far, a chunk that reached its quota is
    closed; when the megabatch does not divide evenly the reference falls back to a strided split (llava_trainer.py:43-44)."""
    if len(indices) % num_chunks != 0:
        return [indices[i::num_chunks] for i in range(num_chunks)]
    quota = len(indices) // num_chunks
    chunks: List[List[int]] = [[] for _ in range(num_chunks)]
    load = [0.0] * num_chunks
    for idx in indices:
        c = load.index(min(load))
        chunks[c].append(idx)
        load[c] += lengths[idx]
        if len(chunks[c]) == quota:
            load[c] = float("inf")
    return chunks


def get_length_grouped_indices(lengths, batch_size: int, world_size: int, generator=None, merge: bool = True) -> List[int]:
    """Random permutation -> megabatches of world_size*batch_size -> each sorted by length (descending, stable) -> split into
    world_size balanced chunks (llava_trainer.py:89-97)."""
    perm = torch.randperm(len(lengths), generator=generator)
    mb = world_size * batch_size
    out: List[int] = []
    for i in range(0, len(lengths), mb):
        mega = sorted(perm[i:i + mb].tolist(), key=lambda j: lengths[j], reverse=True)
        for chunk in split_to_even_chunks(mega, lengths, world_size):
            out.extend(chunk)
    return out


def get_modality_length_grouped_indices(lengths, batch_size: int, world_size: int, generator=None) -> List[int]:
    """lengths > 0 mark multimodal samples, < 0 language-only ones; each family is length-grouped on its own (with the global torch
    RNG: the reference passes generator=None there, :70-71), full megabatches are shuffled together with `generator`, and the two
    incomplete tails form one last sorted batch (:60-86)."""
    assert all(l != 0 for l in lengths), "Should not have zero length."
    mm = [(i, l) for i, l in enumerate(lengths) if l > 0]
    lang = [(i, -l) for i, l in enumerate(lengths) if l < 0]
    assert len(mm) > 0, "Should have at least one multimodal sample."
    assert len(lang) > 0, "Should have at least one language sample."
    mm_idx, mm_len = zip(*mm)
    lang_idx, lang_len = zip(*lang)
    mm_shuffle = [mm_idx[i] for i in get_length_grouped_indices(mm_len, batch_size, world_size, generator=None)]
    lang_shuffle = [lang_idx[i] for i in get_length_grouped_indices(lang_len, batch_size, world_size, generator=None)]
    mb = world_size * batch_size
    mm_mega = [mm_shuffle[i:i + mb] for i in range(0, len(mm_shuffle), mb)]
    lang_mega = [lang_shuffle[i:i + mb] for i in range(0, len(lang_shuffle), mb)]
    tail = mm_mega[-1] + lang_mega[-1]
    mega = mm_mega[:-1] + lang_mega[:-1]
    order = torch.randperm(len(mega), generator=generator)
    mega = [mega[i] for i in order]
    if len(tail) > 0:
        mega.append(sorted(tail))
    return [i for m in mega for i in m]


class LengthGroupedSampler(Sampler):
    """llava_trainer.py:133-165."""

    def __init__(self, batch_size: int, world_size: int, lengths: Optional[List[int]] = None, generator=None, group_by_modality: bool = False):
        if lengths is None:
            raise ValueError("Lengths must be provided.")
        self.batch_size = batch_size
        self.world_size = world_size
        self.lengths = lengths
        self.generator = generator
        self.group_by_modality = group_by_modality

    def __len__(self):
        return len(self.lengths)

    def __iter__(self):
        fn = get_modality_length_grouped_indices if self.group_by_modality else get_length_grouped_indices
        return iter(fn(self.lengths, self.batch_size, self.world_size, generator=self.generator))
