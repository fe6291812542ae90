"""Oracle (TEST INFRASTRUCTURE): beam search as transformers==4.31 runs it for `model.generate(num_beams=k)` - the call the reference's eval
loader forwards (modelcompose/eval/model_multimodal_qa_loader.py:94-102, --num_beams).  The algorithm lives in a third-party dependency
absent from /root/reference (transformers 4.31: generation/utils.py `beam_search`, generation/beam_search.py `BeamSearchScorer` /
`BeamHypotheses`); it is restated here from its published behaviour and pinned against the installed transformers' beam search on a tiny
Llama where the two releases agree (tests/test_beam_cpu.py) - "parity unpinned by the reference" for this piece, as for the sampling warpers.

Restated behaviour (defaults of GenerationConfig: length_penalty 1.0, early_stopping False, num_return_sequences 1, no logits processors):
  * every prompt is repeated k times; beam scores start at (0, -1e9, ..., -1e9) so that the first step expands beam 0 only;
  * a step: log_softmax of the last position's logits + the beam's score; per prompt the 2k best (score, beam, token) candidates in
    descending order (torch.topk over the k * V flattened scores: ties -> lower flat index);
  * candidates are visited in rank order: an EOS candidate among the first k ranks closes a hypothesis (the beam's ids WITHOUT the EOS,
    score = sum_logprobs / len(ids) ** length_penalty with len counting the prompt - 4.31; later releases divide by the generated length),
    an EOS candidate of rank >= k is skipped, any other candidate continues a beam until k beams are filled;
  * a prompt is done when it holds k hypotheses and the worst of them scores at least best_running_sum / cur_len ** length_penalty
    (early_stopping False heuristic; cur_len = len(ids) + 1: BeamSearchScorer.process "adds up to the length which the next_scores is
    calculated on", i.e. the prompt and the step's new token - the 4.31 source is not in this image, the installed 5.x release has another
    stopping rule, so this detail is pinned by tests/test_beam_cpu.py::test_done_heuristic_counts_the_new_token only as a stated choice);
  * at the end (all prompts done, or max length) the running beams of prompts that are not done are added as hypotheses, and the best
    hypothesis per prompt is returned, right-padded with pad_token_id; a hypothesis closed by EOS gets the EOS back if there is room."""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch


class _Hyps:
    def __init__(self, k: int, length_penalty: float, early_stopping: bool):
        self.k, self.lp, self.early = k, length_penalty, early_stopping
        self.beams: List[Tuple[float, torch.Tensor]] = []
        self.worst = 1e9

    def add(self, ids: torch.Tensor, sum_logprobs: float):
        score = sum_logprobs / (ids.shape[-1] ** self.lp)
        if len(self.beams) < self.k or score > self.worst:
            self.beams.append((score, ids))
            if len(self.beams) > self.k:
                srt = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[srt[0][1]]
                self.worst = srt[1][0]
            else:
                self.worst = min(score, self.worst)

    def is_done(self, best_sum_logprobs: float, cur_len: int) -> bool:
        if len(self.beams) < self.k:
            return False
        if self.early:
            return True
        return self.worst >= best_sum_logprobs / cur_len ** self.lp


def beam_search(last_logits: Callable[[torch.Tensor], torch.Tensor], input_ids: torch.Tensor, num_beams: int, max_new_tokens: int,
                eos_token_id: Optional[int], pad_token_id: int, length_penalty: float = 1.0, early_stopping: bool = False,
                _done_len_offset: int = 1) -> torch.Tensor:
    """last_logits(ids (B k, L)) -> (B k, V) fp32 logits of the last position (no cache: the oracle recomputes the prefix).
    Returns (B, <= L + max_new_tokens) int64, the prompt followed by the best hypothesis, padded with pad_token_id."""
    B, L0 = input_ids.shape
    k = num_beams
    ids = input_ids.repeat_interleave(k, dim=0)
    scores = torch.zeros(B, k, dtype=torch.float32)
    scores[:, 1:] = -1e9
    scores = scores.view(-1)
    hyps = [_Hyps(k, length_penalty, early_stopping) for _ in range(B)]
    done = [False] * B
    max_len = L0 + max_new_tokens
    while True:
        cur_len = ids.shape[1]
        logp = torch.log_softmax(last_logits(ids).float(), dim=-1)
        V = logp.shape[-1]
        cand = (logp + scores[:, None]).view(B, k * V)
        top_s, top_i = torch.topk(cand, 2 * k, dim=1, largest=True, sorted=True)
        top_beam, top_tok = top_i // V, top_i % V
        nxt_scores = torch.zeros(B, k)
        nxt_tok = torch.zeros(B, k, dtype=torch.long)
        nxt_idx = torch.zeros(B, k, dtype=torch.long)
        for b in range(B):
            if done[b]:
                nxt_scores[b], nxt_tok[b], nxt_idx[b] = 0.0, pad_token_id, b * k      # padded beams of a finished prompt
                continue
            n = 0
            for rank in range(2 * k):
                tok, sc, src = int(top_tok[b, rank]), float(top_s[b, rank]), b * k + int(top_beam[b, rank])
                if eos_token_id is not None and tok == eos_token_id:
                    if rank >= k:
                        continue
                    hyps[b].add(ids[src].clone(), sc)
                else:
                    nxt_scores[b, n], nxt_tok[b, n], nxt_idx[b, n] = sc, tok, src
                    n += 1
                if n == k:
                    break
            done[b] = done[b] or hyps[b].is_done(float(top_s[b].max()), cur_len + _done_len_offset)
        scores = nxt_scores.view(-1)
        ids = torch.cat([ids[nxt_idx.view(-1)], nxt_tok.view(-1, 1)], dim=1)
        if all(done) or ids.shape[1] >= max_len:
            break
    for b in range(B):
        if done[b]:
            continue
        for j in range(k):
            hyps[b].add(ids[b * k + j], float(scores[b * k + j]))
    best = [sorted(h.beams, key=lambda t: t[0])[-1][1] for h in hyps]           # (sorted(...).pop() of BeamSearchScorer.finalize)
    out_len = min(max(int(x.shape[0]) for x in best) + 1, max_len)
    out = torch.full((B, out_len), pad_token_id, dtype=torch.long)
    for b, x in enumerate(best):
        out[b, :x.shape[0]] = x
        if eos_token_id is not None and x.shape[0] < out_len:                    # "fill with hypotheses and eos_token_id if the latter fits in"
            out[b, x.shape[0]] = eos_token_id
    return out
