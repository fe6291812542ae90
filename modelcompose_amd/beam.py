"""Hypothesis bookkeeping of beam search - transformers 4.31 generation/beam_search.py `BeamHypotheses` (third-party, absent from the
reference tree; `model.generate(num_beams=k)` of eval/model_multimodal_qa_loader.py:94-102 runs it).  Host logic only: the scores it is fed
come from the device (MultimodalLlamaForCausalLM._beam_search)."""
from __future__ import annotations

from typing import List, Tuple

import torch


class BeamHypotheses:
    """The k best finished hypotheses of one prompt.  score = sum_logprobs / len(ids) ** length_penalty, len counting the prompt (4.31)."""

    def __init__(self, num_beams: int, length_penalty: float = 1.0, early_stopping: bool = False):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.beams: List[Tuple[float, torch.Tensor]] = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, ids: torch.Tensor, sum_logprobs: float):
        score = sum_logprobs / (ids.shape[-1] ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, ids))
            if len(self) > self.num_beams:
                ranked = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int) -> bool:
        """early_stopping False: done once no running beam can still beat the worst kept hypothesis (heuristic of 4.31)."""
        if len(self) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty

    def best(self) -> torch.Tensor:
        return sorted(self.beams, key=lambda t: t[0])[-1][1]
