"""Oracle (TEST INFRASTRUCTURE): modal-token splice, torch CPU.

Restates modelcompose/model/multimodal_arch.py:197-459
(encode_modal_inputs, modal_token_match, prepare_inputs_labels_for_multimodal).
Integer / boolean outputs (labels, attention mask, per-modality masks) must be
bit-exact; embeddings are gathers/concats of fp32 rows and are exact too.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100
# modelcompose/constants.py:23-31
MODAL_TOKEN_INDEXES = {"vision": -200, "relrep": -201, "text": -202, "audio": -203, "video": -204, "point": -205}


def encode_modal_inputs(inputs: Dict[str, object], modals: List[str], encode_fn: Dict[str, Callable],
                        prefix_tokens: Optional[Dict[str, torch.Tensor]], suffix_tokens: Optional[Dict[str, torch.Tensor]],
                        skip_absent: bool = False):
    """multimodal_arch.py:197-268.  ``encode_fn[modal](modal_inputs_or_None)`` returns
    the projected features (n_items, T, hidden) (encoder → projector, incl. the
    video (b,t,n,d)->(b,t*n,d) reshape of :236-240).  Absent modalities run the
    encoder on its dummy input in the reference (:203-206); their outputs are
    never spliced, so ``skip_absent=True`` (what the device path does) is
    results-identical except that the returned dict lacks those keys."""
    feats, masks = {}, {}
    for modal in modals:
        if modal not in inputs and skip_absent:
            continue
        f = encode_fn[modal](inputs.get(modal))
        b = f.shape[0]
        parts = []
        if prefix_tokens is not None and modal in prefix_tokens:
            parts.append(prefix_tokens[modal].expand(b, -1, -1))
        parts.append(f)
        if suffix_tokens is not None and modal in suffix_tokens:
            parts.append(suffix_tokens[modal].expand(b, -1, -1))
        f = torch.cat(parts, dim=1)
        feats[modal] = f
        masks[modal] = torch.ones(b, f.shape[1])
    return feats, masks


def modal_token_match(ids: torch.Tensor):
    """multimodal_arch.py:270-285 (earliest sentinel; start_index sentinel value 10000)."""
    modal, start = None, 10000
    for m, tok in MODAL_TOKEN_INDEXES.items():
        hit = ids == tok
        if hit.sum() != 0:
            s = int(torch.where(hit)[0][0])
            if s < start:
                start, modal = s, m
    return modal, start


def prepare_inputs_labels_for_multimodal(input_ids, attention_mask, labels, modal_inputs_keys: List[str],
                                         modal_features: Dict[str, torch.Tensor],
                                         modal_features_attention_mask: Dict[str, torch.Tensor],
                                         embed_weight: torch.Tensor):
    """multimodal_arch.py:298-459 after encode_modal_inputs.  ``modal_inputs_keys`` are
    the keys of the caller's modal_inputs dict (they decide which per-modality masks
    exist, :317-318), ``modal_features`` may also hold absent modalities (:306-307)."""
    embed = lambda ids: F.embedding(ids, embed_weight)
    new_embeds, new_labels = [], ([] if labels is not None else None)
    cur_idx = {m: 0 for m in MODAL_TOKEN_INDEXES}
    mam: Dict[str, list] = {m: [] for m in modal_features}
    for b, cur in enumerate(input_ids):
        cur_e, cur_m = [], {m: [] for m in modal_inputs_keys}
        if labels is not None:
            cur_l, cur_nl = labels[b], []
        modal, start = modal_token_match(cur)
        if modal is None:                                                        # :323-342
            half = cur.shape[0] // 2
            e = torch.cat([embed(cur[:half]), embed(cur[half:])], dim=0)
            new_embeds.append(e)
            if labels is not None:
                new_labels.append(labels[b])
            for m in modal_features:
                mam[m].append(torch.full((len(e),), False, dtype=attention_mask.dtype))
            continue
        while modal is not None:                                                 # :344-367
            f = modal_features[modal][cur_idx[modal]]
            fm = modal_features_attention_mask[modal][cur_idx[modal]].to(attention_mask.dtype)
            cur_e.append(embed(cur[:start]))
            cur_e.append(f)
            for m in cur_m:
                if m != modal:
                    cur_m[m].append(torch.full((start + len(f),), False, dtype=attention_mask.dtype))
                else:
                    cur_m[m].append(torch.full((start,), False, dtype=attention_mask.dtype))
                    cur_m[m].append(fm)
            if labels is not None:
                cur_nl.append(cur_l[:start])
                cur_nl.append(torch.full((f.shape[0],), IGNORE_INDEX, dtype=labels.dtype))
                cur_l = cur_l[start + 1:]
            cur_idx[modal] += 1
            cur = cur[start + 1:]
            modal, start = modal_token_match(cur)
        if cur.numel() > 0:                                                      # :369-375
            cur_e.append(embed(cur))
            for m in cur_m:
                cur_m[m].append(torch.full((len(cur),), False, dtype=attention_mask.dtype))
            if labels is not None:
                cur_nl.append(cur_l)
        new_embeds.append(torch.cat(cur_e, dim=0))
        for m in cur_m:
            mam[m].append(torch.cat(cur_m[m], dim=0))
        if labels is not None:
            new_labels.append(torch.cat(cur_nl, dim=0))

    if any(x.shape != new_embeds[0].shape for x in new_embeds):                  # :390-430 (ragged)
        max_len = max(x.shape[0] for x in new_embeds)
        lens = [x.shape[0] for x in new_embeds]
        new_embeds = torch.stack([torch.cat((x, torch.zeros((max_len - x.shape[0], x.shape[1]), dtype=x.dtype)), 0)
                                  for x in new_embeds], 0)
        out_m = {}
        for m in mam:
            # reference would torch.stack([]) for an absent modality (Appendix B); we
            # only keep modalities that produced masks
            if len(mam[m]) == 0:
                continue
            out_m[m] = torch.stack([torch.cat((x, torch.full((max_len - x.shape[0],), False, dtype=attention_mask.dtype)), 0)
                                    for x in mam[m]], 0)
        mam = out_m
        if labels is not None:
            raw = new_labels
            new_labels = torch.stack([torch.cat((x, torch.full((max_len - x.shape[0],), IGNORE_INDEX, dtype=x.dtype)), 0)
                                      for x in raw], 0)
            if attention_mask is not None:
                rows = []
                for am, nl, nla in zip(attention_mask, raw, new_labels):
                    left = torch.full((nl.shape[0] - labels.shape[1],), True, dtype=attention_mask.dtype)
                    right = torch.full((nla.shape[0] - nl.shape[0],), False, dtype=attention_mask.dtype)
                    rows.append(torch.cat((left, am, right), 0))
                attention_mask = torch.stack(rows, 0)
        elif attention_mask is not None:
            # The reference only defines this branch when labels are given (:414-430 reads
            # _new_labels, unbound otherwise -> NameError for ragged inference batches).
            # Documented extension with the same rule: left-extend True by the inserted
            # length, right-pad False up to max_len.
            rows = []
            for am, n in zip(attention_mask, lens):
                left = torch.full((n - input_ids.shape[1],), True, dtype=attention_mask.dtype)
                right = torch.full((max_len - n,), False, dtype=attention_mask.dtype)
                rows.append(torch.cat((left, am, right), 0))
            attention_mask = torch.stack(rows, 0)
    else:                                                                       # :431-449
        new_embeds = torch.stack(new_embeds, 0)
        mam = {m: torch.stack(v, 0) for m, v in mam.items() if len(v)}
        if labels is not None:
            new_labels = torch.stack(new_labels, 0)
        if attention_mask is not None:
            left = torch.full((attention_mask.shape[0], new_embeds.shape[1] - input_ids.shape[1]), True,
                              dtype=attention_mask.dtype)
            attention_mask = torch.cat((left, attention_mask), dim=1)
    if len(mam):                                                                 # :452-453
        mam["default"] = torch.sum(torch.stack([mam[k] for k in mam]), dim=0) == 0
    else:
        mam = None
    return attention_mask, new_embeds, new_labels, mam
