"""Host-side splice planning: sentinel ids -> segment list -> index arrays for the device copy kernels.

Integer work restating modelcompose/model/multimodal_arch.py:270-459 (modal_token_match + the per-sample loop of
prepare_inputs_labels_for_multimodal): repeatedly take the earliest sentinel among {-200..-205}, emit the text
before it, then the next unused feature block of that modality ([prefix | features | suffix], :246-253); labels
-100 over inserted blocks; attention mask left-extended with True by the inserted length (:445-449); one boolean
mask per modality present in modal_inputs plus 'default' = none of the others (:452-453).
The plan additionally carries the routed (adapter-grouped) row order used by the HIP path."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from ..constants import IGNORE_INDEX, MODAL_TOKEN_INDEXES

_SENT = {v: k for k, v in MODAL_TOKEN_INDEXES.items()}


@dataclass
class SplicePlan:
    B: int
    L_text: int
    lens: np.ndarray                 # [B] spliced length per sample (text pads included, as the reference's embeds)
    Lmax: int
    # per spliced token, sequence order (b-major): -1 padded to Lmax
    tok_id: np.ndarray               # [B, Lmax] int64 text token id or -1
    src_modal: np.ndarray            # [B, Lmax] int8 index into modal_order (-1 = text / pad)
    src_row: np.ndarray              # [B, Lmax] int32 row in that modality's flattened feature buffer
    labels: Optional[np.ndarray]     # [B, Lmax] int64
    attention_mask: np.ndarray       # [B, Lmax] bool
    modal_masks: Dict[str, np.ndarray]       # incl. 'default'; None-equivalent = {} when no modal blocks exist
    modal_order: List[str] = field(default_factory=list)
    items_used: Dict[str, int] = field(default_factory=dict)
    # device path: attended length per sample = lens minus the run of masked-out text tokens at the end (right padding).  The kernels
    # take a length per sequence, so masks with zeros anywhere else (left padding, holes) are flagged and refused there.
    valid_lens: Optional[np.ndarray] = None
    mask_is_suffix: bool = True


def plan_splice(input_ids: np.ndarray, attention_mask: Optional[np.ndarray], labels: Optional[np.ndarray],
                modal_inputs_keys: Sequence[str], block_len: Dict[str, int], n_items: Dict[str, int]) -> SplicePlan:
    """input_ids [B, L] int64 with sentinels.  block_len[m] = tokens per item incl. prefix/suffix;
    n_items[m] = items available (consumed globally in order of appearance, :302,:345,:365)."""
    B, L = input_ids.shape
    if attention_mask is None:
        attention_mask = np.ones((B, L), dtype=bool)
    modal_order = [m for m in MODAL_TOKEN_INDEXES if m in modal_inputs_keys]
    midx = {m: i for i, m in enumerate(modal_order)}
    cur = {m: 0 for m in MODAL_TOKEN_INDEXES}
    rows = []
    for b in range(B):
        ids = input_ids[b]
        tok, sm, sr, lab = [], [], [], []
        for j in range(L):
            t = int(ids[j])
            m = _SENT.get(t)
            if m is None:
                tok.append(t); sm.append(-1); sr.append(-1)
                if labels is not None:
                    lab.append(int(labels[b, j]))
                continue
            if m not in midx:
                raise ValueError(f"input_ids contain the {m} sentinel ({t}) but modal_inputs has no '{m}' entry")
            if cur[m] >= n_items[m]:
                raise ValueError(f"sample {b} needs item {cur[m]} of modality '{m}' but only {n_items[m]} were given")
            T = block_len[m]
            base = cur[m] * T
            tok.extend([-1] * T); sm.extend([midx[m]] * T); sr.extend(range(base, base + T))
            if labels is not None:
                lab.extend([IGNORE_INDEX] * T)
            cur[m] += 1
        rows.append((tok, sm, sr, lab))
    lens = np.array([len(r[0]) for r in rows], dtype=np.int32)
    Lmax = int(lens.max())
    tok_id = np.full((B, Lmax), -1, dtype=np.int64)
    src_modal = np.full((B, Lmax), -1, dtype=np.int8)
    src_row = np.full((B, Lmax), -1, dtype=np.int32)
    out_labels = np.full((B, Lmax), IGNORE_INDEX, dtype=np.int64) if labels is not None else None
    am = np.zeros((B, Lmax), dtype=bool)
    for b, (tok, sm, sr, lab) in enumerate(rows):
        n = len(tok)
        tok_id[b, :n] = tok; src_modal[b, :n] = sm; src_row[b, :n] = sr
        if labels is not None:
            out_labels[b, :n] = lab
        ins = n - L
        am[b, :ins] = True                               # left-extended by the inserted length
        am[b, ins:n] = attention_mask[b]
    masks: Dict[str, np.ndarray] = {}
    any_block = any(cur[m] > 0 for m in modal_order) or len(modal_order) > 0
    if any_block:
        for m in modal_order:
            masks[m] = src_modal == midx[m]
        acc = np.zeros((B, Lmax), dtype=bool)
        for m in masks:
            acc |= masks[m]
        masks["default"] = ~acc
    n_att = attention_mask.astype(bool).sum(1)
    suffix_only = bool(all(attention_mask[b, :n_att[b]].all() for b in range(B)))
    valid_lens = (lens - (L - n_att)).astype(np.int32)
    return SplicePlan(B, L, lens, Lmax, tok_id, src_modal, src_row, out_labels, am, masks, modal_order,
                      {m: cur[m] for m in modal_order}, valid_lens, suffix_only)


@dataclass
class RoutedLayout:
    M: int
    order_b: np.ndarray          # [M] batch entry of routed row r
    order_t: np.ndarray          # [M] position of routed row r
    group_start: np.ndarray      # [G+1]
    group_adapter: np.ndarray    # [G]
    out_map: np.ndarray          # [B*Lmax] sequence slot -> routed row (-1 = padding)
    last_rows: np.ndarray        # [B]


def routed_layout(plan: SplicePlan, adapter_of_modal: Dict[str, int], routed: bool) -> RoutedLayout:
    """Group valid tokens by adapter (stable in (b, t) order inside a group)."""
    B, Lmax = plan.tok_id.shape
    lens = plan.lens if plan.valid_lens is None else plan.valid_lens
    valid = np.arange(Lmax)[None, :] < lens[:, None]
    adapter = np.zeros((B, Lmax), dtype=np.int32)
    if routed:
        for i, m in enumerate(plan.modal_order):
            adapter[plan.src_modal == i] = adapter_of_modal[m]
    bb, tt = np.nonzero(valid)
    ad = adapter[bb, tt]
    order = np.argsort(ad, kind="stable")
    bb, tt, ad = bb[order], tt[order], ad[order]
    M = len(bb)
    groups = sorted(set(ad.tolist()))
    starts = [int(np.searchsorted(ad, g, side="left")) for g in groups] + [M]
    out_map = np.full(B * Lmax, -1, dtype=np.int32)
    out_map[bb * Lmax + tt] = np.arange(M, dtype=np.int32)
    last = out_map[np.arange(B) * Lmax + (lens - 1)]
    return RoutedLayout(M, bb.astype(np.int32), tt.astype(np.int32), np.array(starts, dtype=np.int32),
                        np.array(groups, dtype=np.int32), out_map, last.astype(np.int32))
