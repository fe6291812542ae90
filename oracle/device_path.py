"""Oracle (TEST INFRASTRUCTURE): the composed-Vicuna backbone with the HIP path's rounding points, torch CPU fp32.

`oracle/llm.py` restates the REFERENCE (branch form, fp32 / optional storage rounding).  This file restates what
`libmc_hip.so` computes for the same function, so that device-vs-oracle comparisons can be held to the tolerance
BASELINE.json names (1e-3 of the logit scale, greedy ids bit-exact) instead of to "one bf16 rounding per op":

  * weights: one dense matrix per routed adapter, W' = bf16((W + sum_m s_m B_m A_m) diag(g)) with the preceding RMSNorm's
    weight g folded into q|k|v and gate|up (csrc/compose.hip; reference: LocalLoraLinear.forward
    modelcompose/model/language_model/multimodal_llama.py:130-157 + LlamaRMSNorm :405-406, dense form
    scripts/evaluate_delta_weights.py:8-15);
  * every token multiplies against the weight of ITS adapter only (one-hot modal masks, multimodal_arch.py:452-453;
    the reference's stack-mask-sum :262-268 adds exact zeros);
  * RMSNorm = per-row fp32 factor applied to the fp32 accumulator (GEMM epilogue), the normalised activations are never rounded;
  * values are rounded to bf16 exactly where the kernels store them: q|k|v, rotated q / k, attention output, both residual
    sums, gate and up (before SiLU), silu(gate)*up, the final norm output;
  * prefill attention: P = bf16(exp(s - rowmax)) multiplies V, the row sum that normalises is taken over the fp32 P (csrc/attention.hip);
    decode attention: fp32 softmax, no rounding of P (attn_decode_kernel);
  * decode steps use the 'default' adapter only (multimodal_llama.py:435-438).
Every rounding point can be switched off one at a time (`rounding={...}`, names in ROUNDING_POINTS): the error budget of
profiles/r04_parity.json and the fp32-residual-stream A/B come from that.  `lazy=True` composes a layer's weights when the layer
runs (a 32-layer model with 4 routed adapters is 104 GB as fp32 tensors; lazily it is the bf16 state dict + one layer).
What is NOT reproduced: fp32 summation order inside the kernels (MFMA tiling, split reductions) and the online-softmax's
running maximum (P is rounded relative to the final row maximum here).  Both perturb results far below one bf16 step.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import llm

BF = torch.bfloat16


def bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(BF).to(torch.float32)


# where the HIP path stores bf16 (True = rounded there, the shipped path).  "operand": the x operand of a GEMM is bf16 even when the
# residual stream it comes from is kept in fp32 ("resid_attn" / "resid_mlp" False = an fp32 residual stream).
ROUNDING_POINTS = ("weights", "embed", "qkv", "rope", "p", "attn_out", "resid_attn", "resid_mlp", "operand", "gate_up", "inter", "final")


def _rounding(r: Optional[dict]) -> dict:
    out = {k: True for k in ROUNDING_POINTS}
    if r:
        unknown = set(r) - set(ROUNDING_POINTS)
        assert not unknown, unknown
        out.update(r)
    return out


class DeviceWeights:
    """Per layer and routed adapter: q|k|v (norm folded), o, gate / up (norm folded), down as fp32 tensors holding bf16 values.
    Adapters without any LoRA term share the base tensors (as the device path shares their storage)."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: llm.LLMConfig, lazy: bool = False, round_weights: bool = True):
        self.cfg, self.sd, self.lazy = cfg, sd, lazy
        self._rw = bf if round_weights else (lambda t: t)
        self._default_cache: Dict[int, Dict[str, torch.Tensor]] = {}      # lazy mode: the decode adapter's layers, kept as bf16 tensors
        self.layers: List[Dict[str, Dict[str, torch.Tensor]]] = []
        if not lazy:
            for i in range(cfg.num_hidden_layers):
                self.layers.append(self._build_layer(i, list(cfg.modal_names)))
        self.final_norm = sd["model.norm.weight"].float()
        self.lm_head = self._rw(sd["lm_head.weight"].float())
        self.embed = self._rw(sd["model.embed_tokens.weight"].float())

    def _build_layer(self, i: int, adapters) -> Dict[str, Dict[str, torch.Tensor]]:
        sd, cfg = self.sd, self.cfg
        p = f"model.layers.{i}"
        g_in = sd[f"{p}.input_layernorm.weight"].float()
        g_post = sd[f"{p}.post_attention_layernorm.weight"].float()
        per, cache = {}, {}
        for ad in adapters:
            w = {}
            for blk, lins in (("self_attn", llm.LINEARS_ATTN), ("mlp", llm.LINEARS_MLP)):
                for lin in lins:
                    pre = f"{p}.{blk}.{lin}"
                    key = (lin, self._terms_key(sd, pre, cfg, ad))
                    if key not in cache:
                        m = llm.merged_weight(sd, pre, cfg, ad)
                        if lin in ("q_proj", "k_proj", "v_proj"):
                            m = m * g_in[None, :]
                        elif lin in ("gate_proj", "up_proj"):
                            m = m * g_post[None, :]
                        cache[key] = self._rw(m)
                    w[lin] = cache[key]
            per[ad] = w
        return per

    def layer(self, i: int, adapters) -> Dict[str, Dict[str, torch.Tensor]]:
        """The dense weights of layer i for the named routed adapters."""
        if not self.lazy:
            return self.layers[i]
        adapters = list(adapters)
        if adapters == ["default"]:
            if i not in self._default_cache:
                w = self._build_layer(i, adapters)["default"]
                # bf16-valued when rounded; an unrounded budget run keeps fp32 (small models only)
                self._default_cache[i] = {k: (v.to(BF) if self._rw is bf else v) for k, v in w.items()}
            return {"default": {k: v.float() for k, v in self._default_cache[i].items()}}
        return self._build_layer(i, adapters)

    @staticmethod
    def _terms_key(sd, pre, cfg, ad):
        names, scaling, default_names, merge = llm.adapter_plan(cfg)
        if f"{pre}.lora_A.{ad}.weight" not in sd or ad not in names:
            return "base"
        return ad


def _routed(x2d: torch.Tensor, weights: Dict[str, Dict[str, torch.Tensor]], lin: str, groups) -> torch.Tensor:
    """x2d [M, K]; groups = [(adapter_name, row_index_tensor)]: every row against its adapter's dense weight, fp32 accumulate."""
    if len(groups) == 1:
        return F.linear(x2d, weights[groups[0][0]][lin])
    out = None
    for ad, rows in groups:
        y = F.linear(x2d[rows], weights[ad][lin])
        if out is None:
            out = torch.empty(x2d.shape[0], y.shape[1], dtype=torch.float32)
        out[rows] = y
    return out


def _rs(x2d: torch.Tensor, eps: float) -> torch.Tensor:
    return torch.rsqrt(x2d.pow(2).mean(-1, keepdim=True) + eps)


def _rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, rnd: bool = True) -> torch.Tensor:
    """x (B, L, H, D) fp32; cos / sin (B, L, D/2): rope_kv_kernel's pairing (i, i + D/2), rounded to bf16."""
    half = x.shape[-1] // 2
    a, b = x[..., :half], x[..., half:]
    c, s = cos[:, :, None, :], sin[:, :, None, :]
    y = torch.cat([a * c - b * s, b * c + a * s], dim=-1)
    return bf(y) if rnd else y


def _groups_of(adapter_masks, decode: bool, B: int, L: int):
    """[(adapter, row indices or None)]: the routed groups of the flattened (B L) rows."""
    if decode or adapter_masks is None:
        return [("default", None)]
    flat = {k: v.reshape(-1) for k, v in adapter_masks.items()}
    groups = [(k, torch.nonzero(m).squeeze(1)) for k, m in flat.items() if bool(m.any())]
    cover = torch.zeros(B * L, dtype=torch.long)
    for _, rows in groups:
        cover[rows] += 1
    assert bool((cover == 1).all()), "adapter masks must be one-hot per token"
    if len(groups) == 1:
        groups = [(groups[0][0], None)]
    return groups


def _layer(dw: DeviceWeights, i: int, h: torch.Tensor, groups, cos, sin, B: int, L: int, R: dict, past=None, trace: Optional[dict] = None):
    """One decoder layer over h (B L, hidden) with the HIP path's storage points (MultimodalLlamaDecoderLayer.forward,
    multimodal_llama.py:408-468).  past: (k, v) of this layer or None.  Returns (h_out, (k, v))."""
    cfg = dw.cfg
    r = lambda name, t: bf(t) if R[name] else t
    opx = (lambda t: t) if (R["resid_attn"] and R["resid_mlp"]) or not R["operand"] else bf      # GEMM operand of an fp32 stream
    H, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    eps = cfg.rms_norm_eps
    decode = past is not None
    scale = 1.0 / math.sqrt(D)
    W = dw.layer(i, [g_[0] for g_ in groups])
    rs = _rs(h, eps)
    hx = opx(h)
    q = r("qkv", _routed(hx, W, "q_proj", groups) * rs).view(B, L, H, D)
    k = r("qkv", _routed(hx, W, "k_proj", groups) * rs).view(B, L, Hkv, D)
    v = r("qkv", _routed(hx, W, "v_proj", groups) * rs).view(B, L, Hkv, D)
    if trace is not None:
        trace[f"{i}.rs"] = rs.clone()
        trace[f"{i}.qkv"] = torch.cat([q.reshape(B * L, -1), k.reshape(B * L, -1), v.reshape(B * L, -1)], 1)
    q, k = _rope(q, cos, sin, R["rope"]), _rope(k, cos, sin, R["rope"])
    if trace is not None:
        trace[f"{i}.q_rot"] = q.reshape(B * L, -1).clone()
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)            # (B, H, L, D)
    if decode:
        k = torch.cat([past[0], k], dim=2)
        v = torch.cat([past[1], v], dim=2)
    present = (k, v)
    rep = H // Hkv
    kk = k if rep == 1 else k.repeat_interleave(rep, dim=1)
    vv = v if rep == 1 else v.repeat_interleave(rep, dim=1)
    s = torch.matmul(q, kk.transpose(2, 3)) * scale
    if decode:
        p = torch.softmax(s, dim=-1)
        o = torch.matmul(p, vv)
    else:
        causal = torch.ones(L, L, dtype=torch.bool).tril()
        s = s.masked_fill(~causal, float("-inf"))
        m = s.max(dim=-1, keepdim=True).values
        p = torch.exp(s - m)
        o = torch.matmul(r("p", p), vv) / p.sum(-1, keepdim=True)
    o = r("attn_out", o).transpose(1, 2).reshape(B * L, H * D)
    h = r("resid_attn", h + _routed(o, W, "o_proj", groups))
    if trace is not None:
        trace[f"{i}.attn"] = o.clone()
        trace[f"{i}.x1"] = h.clone()
    rs = _rs(h, eps)
    hx = opx(h)
    g = r("gate_up", _routed(hx, W, "gate_proj", groups) * rs)
    u = r("gate_up", _routed(hx, W, "up_proj", groups) * rs)
    inter = r("inter", g / (1.0 + torch.exp(-g)) * u)
    h = r("resid_mlp", h + _routed(inter, W, "down_proj", groups))
    del W
    if trace is not None:
        trace[f"{i}.inter"] = inter.clone()
        trace[f"{i}.x2"] = h.clone()
    return h, present


def _cos_sin(cfg, B: int, L: int, past_len: int):
    D = cfg.head_dim
    cos_t, sin_t = llm.rope_tables(D, max(cfg.max_position_embeddings, past_len + L), cfg.rope_theta)
    half = D // 2
    pos = torch.arange(past_len, past_len + L)
    return cos_t[pos][:, :half][None].expand(B, L, half), sin_t[pos][:, :half][None].expand(B, L, half)


def forward_layer(dw: DeviceWeights, i: int, x: torch.Tensor, adapter_masks: Optional[Dict[str, torch.Tensor]], rounding: Optional[dict] = None,
                  trace: Optional[dict] = None) -> torch.Tensor:
    """Prefill of decoder layer i ALONE on a given input x (B, L, hidden) fp32 holding bf16 values - the hidden state that enters the layer -
    returning the layer's output (B, L, hidden).  For per-layer teacher-forced checks at depth (tests/test_fulldepth_parity_gpu.py): both
    implementations are handed the same input of layer i, so their outputs differ by one layer's worth of rounding, not by i layers'."""
    B, L, Hd = x.shape
    groups = _groups_of(adapter_masks, False, B, L)
    cos, sin = _cos_sin(dw.cfg, B, L, 0)
    h, _ = _layer(dw, i, x.reshape(B * L, Hd), groups, cos, sin, B, L, _rounding(rounding), None, trace)
    return h.view(B, L, Hd)


def forward(dw: DeviceWeights, x: torch.Tensor, adapter_masks: Optional[Dict[str, torch.Tensor]], past_kv=None, last_only=False,
            trace: Optional[dict] = None, rounding: Optional[dict] = None):
    """x (B, L, hidden) fp32 holding bf16 values (spliced embeddings, or the embedding rows of one decode token).
    adapter_masks: {adapter: bool (B, L)} one-hot over the routed adapters, or None (= every token 'default').
    Returns (logits fp32 (B, L or 1, vocab), present_kv)."""
    cfg = dw.cfg
    R = _rounding(rounding)
    r = lambda name, t: bf(t) if R[name] else t
    B, L, Hd = x.shape
    eps = cfg.rms_norm_eps
    decode = past_kv is not None
    past_len = past_kv[0][0].shape[2] if decode else 0
    groups = _groups_of(adapter_masks, decode, B, L)
    cos, sin = _cos_sin(cfg, B, L, past_len)
    h = x.reshape(B * L, Hd)
    presents = []
    for i in range(cfg.num_hidden_layers):
        h, present = _layer(dw, i, h, groups, cos, sin, B, L, R, past_kv[i] if decode else None, trace)
        presents.append(present)
    hid = h.view(B, L, Hd)
    if last_only:
        hid = hid[:, -1:]
    nl = r("final", hid * _rs(hid, eps) * dw.final_norm)
    return F.linear(nl, dw.lm_head), tuple(presents)
