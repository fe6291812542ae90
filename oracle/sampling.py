"""Oracle (TEST INFRASTRUCTURE): the sampled next-token rule behind generate(do_sample=True, temperature, top_p) —
modelcompose/eval/model_multimodal_qa_loader.py:94-102 (default --temperature 0.2), modelcompose/serve/model_worker.py:160-185.

The arithmetic lives in a third-party dependency that is absent from /root/reference: transformers==4.31.0 (requirements pin),
generation/logits_process.py (TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper) and generation/utils.py `sample`
(softmax + torch.multinomial).  Restated here from the published algorithm; tests/test_oracle_golden.py pins it against the warper
classes of the transformers installed in this image (same definitions as 4.31 for these three)."""
from __future__ import annotations

import torch


def warp(logits: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> torch.Tensor:
    """logits [M, V] fp32 -> warped scores (removed tokens = -inf), in generate()'s warper order."""
    s = logits / temperature                                                # TemperatureLogitsWarper
    if top_k and top_k > 0:                                                 # TopKLogitsWarper: ties at the k-th value stay
        k = min(top_k, s.shape[-1])
        s = s.masked_fill(s < torch.topk(s, k)[0][..., -1, None], -float("inf"))
    if top_p is not None and top_p < 1.0:                                   # TopPLogitsWarper, min_tokens_to_keep = 1
        sorted_logits, sorted_idx = torch.sort(s, descending=False)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[..., -1:] = False
        s = s.masked_fill(remove.scatter(1, sorted_idx, remove), -float("inf"))
    return s


def probabilities(logits, temperature=1.0, top_k=0, top_p=1.0) -> torch.Tensor:
    return torch.softmax(warp(logits, temperature, top_k, top_p), dim=-1)


def pick(probs: torch.Tensor, uniform: torch.Tensor) -> torch.Tensor:
    """Inverse-CDF draw in index order: first index whose inclusive cumulative probability exceeds u * total (float64)."""
    c = probs.double().cumsum(dim=-1)
    t = uniform.double()[:, None] * c[:, -1:]
    return (c > t).float().argmax(dim=-1)
