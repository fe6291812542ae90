"""Oracle (TEST INFRASTRUCTURE): composed-Vicuna language model, torch CPU fp32.

Restates, function by function, the reference files
  modelcompose/model/language_model/multimodal_llama.py   (LocalLoRA llama)
and the transformers==4.31 llama helpers it star-imports (RMSNorm, rotary
tables, rotate-half RoPE, causal+padding additive mask), which are absent from
/root/reference and therefore restated from their published behaviour.

Weights are a flat ``dict[str, Tensor]`` that uses the reference's own
state_dict key grammar (SURVEY.md §5 "checkpoint"), so reference checkpoints /
golden fixtures load without renaming.

``emulate`` selects where values are rounded:
  None    – pure fp32, the branch form of the reference (used for the golden pin)
  "bf16"  – fp32 arithmetic with bf16 rounding at the device path's storage
            points and *pre-merged* bf16 weights (what the HIP path computes);
            used for tight device-vs-oracle comparisons and exact greedy ids.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

LINEARS_ATTN = ("q_proj", "k_proj", "v_proj", "o_proj")
LINEARS_MLP = ("gate_proj", "up_proj", "down_proj")


@dataclass
class LLMConfig:
    vocab_size: int = 32000
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    max_position_embeddings: int = 4096
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    lora_r: int = 128
    lora_alpha: int = 256
    lora_strategy: Optional[str] = "modal+language"
    # adapter order = infer_modals: default, audio, vision, video, point
    # (multimodal_encoder/builder.py:119-129)
    modal_names: Sequence[str] = ("default",)
    reset_scaling_weights: Optional[str] = None
    merge_default_weights: Optional[str] = None
    pad_token_id: Optional[int] = 0
    eos_token_id: int = 2

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def rnd(x: torch.Tensor, emulate: Optional[str]) -> torch.Tensor:
    if emulate is None:
        return x
    if emulate == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if emulate == "fp16":
        return x.to(torch.float16).to(torch.float32)
    raise ValueError(emulate)


# ----------------------------------------------------------------------------
# LocalLoraLinear  (multimodal_llama.py:68-160)
# ----------------------------------------------------------------------------
def extract_params(s: str) -> Dict[str, float]:
    """multimodal_llama.py:109-118."""
    out = {}
    for pair in s.split(","):
        k, v = pair.split("=")
        out[k.strip()] = float(v)
    return out


def adapter_plan(cfg: LLMConfig) -> Tuple[List[str], Dict[str, float], Optional[List[str]], Optional[str]]:
    """Adapter set, scaling dict, default_adapter_names, merge mode.

    Follows LocalLoraLinear.__init__ (multimodal_llama.py:84-107): adapters are
    created for every name in modal_names; when any key of reset_scaling_weights
    contains 'default-', a 'default-{m}' adapter is created for EVERY non-default
    modality and merge_default_weights is forced to 'linear-'; scaling[k] =
    alpha/r, multiplied by the coefficient for keys named in the string.
    """
    names = list(cfg.modal_names)
    scaling = {n: cfg.lora_alpha / cfg.lora_r for n in names}
    merge = cfg.merge_default_weights
    default_names = None
    if cfg.reset_scaling_weights is not None:
        reset = extract_params(cfg.reset_scaling_weights)
        if any("default-" in k for k in reset):
            merge = "linear-"
            default_names = [f"default-{n}" for n in names[1:]]
            for dn in default_names:
                names.append(dn)
                scaling[dn] = cfg.lora_alpha / cfg.lora_r
        for k in reset:
            if k in scaling:
                scaling[k] = scaling[k] * reset[k]
    return names, scaling, default_names, merge


# training only: nn.Dropout on the LoRA input (multimodal_llama.py:133, :150-153).  DROPOUT_FN(prefix, adapter, x) -> dropped x; None in
# eval mode.  The reference draws an independent mask per adapter module; a token only ever uses its own adapter's branch
# (one-hot routing), so one mask per (linear, token) is the same distribution - oracle/train.py passes explicit masks.
DROPOUT_FN = None


def _lora_branch(x, sd, prefix, adapter, scale):
    a = sd.get(f"{prefix}.lora_A.{adapter}.weight")
    b = sd.get(f"{prefix}.lora_B.{adapter}.weight")
    if DROPOUT_FN is not None:
        x = DROPOUT_FN(prefix, adapter, x)
    return F.linear(F.linear(x, a), b) * scale


def lora_linear(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str, cfg: LLMConfig,
                active_adapters: Optional[Sequence[str]]) -> Dict[str, torch.Tensor]:
    """LocalLoraLinear.forward (multimodal_llama.py:120-160), eval mode (dropout = identity)."""
    names, scaling, default_names, merge = adapter_plan(cfg)
    original = F.linear(x, sd[f"{prefix}.weight"])
    if not active_adapters:
        return original
    outputs = {}
    for ad in active_adapters:
        if f"{prefix}.lora_A.{ad}.weight" not in sd or ad not in names:
            outputs[ad] = original                                     # :127-129
            continue
        if ad == "default" and merge is not None:                     # :130-149
            subs = [_lora_branch(x, sd, prefix, dn, scaling[dn]) for dn in default_names]
            if merge == "sum" or merge.startswith("linear-"):
                branch = torch.stack(subs).sum(0)
            elif merge == "mean":
                branch = torch.stack(subs).mean(0)
            else:
                raise NotImplementedError(f"online merging strategy '{merge}' is not implemented.")
            outputs["default"] = original + branch
            continue
        outputs[ad] = original + _lora_branch(x, sd, prefix, ad, scaling[ad])   # :150-157
    return outputs


def dense_delta(sd, prefix, adapter, scale) -> torch.Tensor:
    """ΔW = (B @ A)·scale  (scripts/evaluate_delta_weights.py:8-15)."""
    return (sd[f"{prefix}.lora_B.{adapter}.weight"].float() @ sd[f"{prefix}.lora_A.{adapter}.weight"].float()) * scale


def merged_weight(sd, prefix, cfg: LLMConfig, adapter: str) -> torch.Tensor:
    """Dense weight that reproduces lora_linear(...)[adapter]:  W + Σ scale·B·A  (fp32)."""
    names, scaling, default_names, merge = adapter_plan(cfg)
    w = sd[f"{prefix}.weight"].float()
    if f"{prefix}.lora_A.{adapter}.weight" not in sd or adapter not in names:
        return w.clone()
    if adapter == "default" and merge is not None:
        acc = torch.zeros_like(w)
        for dn in default_names:
            acc = acc + dense_delta(sd, prefix, dn, scaling[dn])
        if merge == "mean":
            acc = acc / len(default_names)
        return w + acc
    return w + dense_delta(sd, prefix, adapter, scaling[adapter])


# ----------------------------------------------------------------------------
# transformers==4.31 helpers (third-party, restated)
# ----------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """LlamaRMSNorm 4.31: fp32 variance, cast back, then multiply by weight."""
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return weight * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def rope_tables(head_dim: int, n_pos: int, theta: float = 10000.0):
    """LlamaRotaryEmbedding 4.31: inv_freq fp32, emb = cat(freqs, freqs)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(n_pos, dtype=torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin, position_ids):
    """apply_rotary_pos_emb 4.31 (gather by position, rotate-half pairs (i, i+d/2))."""
    c = cos[position_ids].unsqueeze(1)   # (B,1,L,D)
    s = sin[position_ids].unsqueeze(1)
    return q * c + rotate_half(q) * s, k * c + rotate_half(k) * s


def decoder_attention_mask(attention_mask: Optional[torch.Tensor], bsz: int, q_len: int, past_len: int,
                           dtype=torch.float32) -> Optional[torch.Tensor]:
    """LlamaModel._prepare_decoder_attention_mask 4.31: causal finfo.min triangle
    (zero columns for the past) PLUS the expanded padding mask (finfo.min where 0)."""
    neg = torch.finfo(dtype).min
    combined = None
    if q_len > 1:
        m = torch.full((q_len, q_len), neg, dtype=dtype)
        idx = torch.arange(q_len)
        m.masked_fill_(idx < (idx + 1).view(q_len, 1), 0)
        if past_len > 0:
            m = torch.cat([torch.zeros(q_len, past_len, dtype=dtype), m], dim=-1)
        combined = m[None, None].expand(bsz, 1, q_len, q_len + past_len)
    if attention_mask is not None:
        am = attention_mask[:, None, None, :].expand(bsz, 1, q_len, attention_mask.shape[-1]).to(dtype)
        inv = 1.0 - am
        exp = inv.masked_fill(inv.to(torch.bool), neg)
        combined = exp if combined is None else exp + combined
    return combined


# ----------------------------------------------------------------------------
# Attention / MLP / layer / model  (multimodal_llama.py:162-619)
# ----------------------------------------------------------------------------
def _route(mapping: Dict[str, torch.Tensor], modal_mask: Dict[str, torch.Tensor], like: torch.Tensor):
    """stack([out[k] * mask[k]]).sum(0) over the keys of modal_attention_mask (:266-268, :336, :390)."""
    return torch.stack([mapping[k] * modal_mask[k].unsqueeze(-1).to(like) for k in modal_mask]).sum(dim=0)


def attention(h, sd, pre, cfg: LLMConfig, attn_mask, modal_mask, position_ids, past_kv, cos, sin, emulate=None):
    """LocalLoraAttention.forward (multimodal_llama.py:210-342)."""
    bsz, q_len, _ = h.shape
    H, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    if modal_mask is None:                                                          # :239-242
        q = lora_linear(h, sd, f"{pre}.q_proj", cfg, ("default",))["default"]
        k = lora_linear(h, sd, f"{pre}.k_proj", cfg, ("default",))["default"]
        v = lora_linear(h, sd, f"{pre}.v_proj", cfg, ("default",))["default"]
    else:                                                                           # :262-268
        q = _route(lora_linear(h, sd, f"{pre}.q_proj", cfg, cfg.modal_names), modal_mask, h)
        k = _route(lora_linear(h, sd, f"{pre}.k_proj", cfg, cfg.modal_names), modal_mask, h)
        v = _route(lora_linear(h, sd, f"{pre}.v_proj", cfg, cfg.modal_names), modal_mask, h)
    q, k, v = rnd(q, emulate), rnd(k, emulate), rnd(v, emulate)
    q = q.view(bsz, q_len, H, D).transpose(1, 2)
    k = k.view(bsz, q_len, Hkv, D).transpose(1, 2)
    v = v.view(bsz, q_len, Hkv, D).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin, position_ids)                                 # :281-282
    q, k = rnd(q, emulate), rnd(k, emulate)
    if past_kv is not None:                                                         # :284-287
        k = torch.cat([past_kv[0], k], dim=2)
        v = torch.cat([past_kv[1], v], dim=2)
    present = (k, v)
    rep = H // Hkv
    kk = k if rep == 1 else k[:, :, None].expand(bsz, Hkv, rep, k.shape[2], D).reshape(bsz, H, k.shape[2], D)
    vv = v if rep == 1 else v[:, :, None].expand(bsz, Hkv, rep, v.shape[2], D).reshape(bsz, H, v.shape[2], D)
    w = torch.matmul(q, kk.transpose(2, 3)) / math.sqrt(D)                          # :295
    if attn_mask is not None:
        w = w + attn_mask                                                           # :308
    w = F.softmax(w, dim=-1, dtype=torch.float32)                                   # :311
    if emulate is not None:
        # device path: P is rounded to the storage dtype before P·V, the row sum
        # stays fp32 (flash-style normalisation after the product).
        m = w.max(dim=-1, keepdim=True).values.clamp_min(1e-30)
        # (p / rowmax) is what the kernel exponentiates against the running max
        p_un = rnd(w / m, emulate)
        o = torch.matmul(p_un, vv) * m
    else:
        o = torch.matmul(w, vv)                                                     # :312
    o = rnd(o, emulate)
    o = o.transpose(1, 2).contiguous().reshape(bsz, q_len, H * D)
    if modal_mask is None:                                                          # :328-336
        o = lora_linear(o, sd, f"{pre}.o_proj", cfg, ("default",))["default"]
    else:
        o = _route(lora_linear(o, sd, f"{pre}.o_proj", cfg, cfg.modal_names), modal_mask, h)
    return o, present


def mlp(x, sd, pre, cfg: LLMConfig, modal_mask, emulate=None):
    """LocalLoraMLP.forward (multimodal_llama.py:363-396); down_proj is applied per
    adapter to that adapter's own act(gate)*up (:381-388)."""
    if modal_mask is not None:
        g = lora_linear(x, sd, f"{pre}.gate_proj", cfg, cfg.modal_names)
        u = lora_linear(x, sd, f"{pre}.up_proj", cfg, cfg.modal_names)
        d = {}
        for m in u:
            inter = rnd(F.silu(rnd(g[m], emulate)) * rnd(u[m], emulate), emulate)
            d[m] = lora_linear(inter, sd, f"{pre}.down_proj", cfg, [m])[m]
        return _route(d, modal_mask, x)
    g = lora_linear(x, sd, f"{pre}.gate_proj", cfg, ("default",))["default"]
    u = lora_linear(x, sd, f"{pre}.up_proj", cfg, ("default",))["default"]
    inter = rnd(F.silu(rnd(g, emulate)) * rnd(u, emulate), emulate)
    return lora_linear(inter, sd, f"{pre}.down_proj", cfg, ("default",))["default"]


def decoder_layer(h, sd, i, cfg, attn_mask, modal_mask, position_ids, past_kv, cos, sin, emulate=None):
    """MultimodalLlamaDecoderLayer.forward (multimodal_llama.py:408-468)."""
    pre = f"model.layers.{i}"
    if past_kv is not None:                                                         # :435-438
        modal_mask = None
    res = h
    n = rnd(rms_norm(h, sd[f"{pre}.input_layernorm.weight"], cfg.rms_norm_eps), emulate)
    a, present = attention(n, sd, f"{pre}.self_attn", cfg, attn_mask, modal_mask, position_ids, past_kv, cos, sin, emulate)
    h = rnd(res + a, emulate)
    res = h
    n = rnd(rms_norm(h, sd[f"{pre}.post_attention_layernorm.weight"], cfg.rms_norm_eps), emulate)
    h = rnd(res + mlp(n, sd, f"{pre}.mlp", cfg, modal_mask, emulate), emulate)
    return h, present


def model_forward(sd, cfg: LLMConfig, inputs_embeds=None, input_ids=None, attention_mask=None,
                  modal_attention_mask=None, past_key_values=None, emulate=None):
    """MultimodalLlamaModel.forward (multimodal_llama.py:488-619). Returns (hidden, kv tuple)."""
    if inputs_embeds is None:
        inputs_embeds = F.embedding(input_ids, sd["model.embed_tokens.weight"])
    bsz, q_len, _ = inputs_embeds.shape
    past_len = 0 if past_key_values is None else past_key_values[0][0].shape[2]
    position_ids = torch.arange(past_len, q_len + past_len, dtype=torch.long).unsqueeze(0).expand(bsz, q_len)  # :526-531
    if attention_mask is None:
        attention_mask = torch.ones((bsz, q_len + past_len), dtype=torch.bool)
    mask4d = decoder_attention_mask(attention_mask, bsz, q_len, past_len)           # :543-545
    cos, sin = rope_tables(cfg.head_dim, max(cfg.max_position_embeddings, q_len + past_len), cfg.rope_theta)
    h = rnd(inputs_embeds.float(), emulate)
    presents = []
    for i in range(cfg.num_hidden_layers):
        pkv = None if past_key_values is None else past_key_values[i]
        h, present = decoder_layer(h, sd, i, cfg, mask4d, modal_attention_mask, position_ids, pkv, cos, sin, emulate)
        presents.append(present)
    h = rnd(rms_norm(h, sd["model.norm.weight"], cfg.rms_norm_eps), emulate)         # :603
    return h, tuple(presents)


def lm_logits(h, sd):
    """lm_head (multimodal_llama.py:720)."""
    return F.linear(h, sd["lm_head.weight"])


def cross_entropy_shifted(logits, labels, vocab):
    """multimodal_llama.py:722-733."""
    sl = logits[..., :-1, :].contiguous().view(-1, vocab)
    tl = labels[..., 1:].contiguous().view(-1)
    return F.cross_entropy(sl, tl, ignore_index=-100)


def premerge_state_dict(sd: Dict[str, torch.Tensor], cfg: LLMConfig, emulate: Optional[str] = "bf16"):
    """Device-path weight preparation, restated on the CPU: for every LocalLoRA
    linear build one dense weight per routed adapter (W + Σ scale·B·A in fp32,
    rounded once to the storage dtype).  Returns {adapter: state-dict-like} where
    each entry only holds '...weight' keys (no lora_*), so the same forward code
    runs with cfg.modal_names=('default',) and no reset string."""
    out = {}
    for ad in cfg.modal_names:
        d = {}
        for k, v in sd.items():
            if ".lora_A." in k or ".lora_B." in k:
                continue
            d[k] = v
        for i in range(cfg.num_hidden_layers):
            for blk, lins in (("self_attn", LINEARS_ATTN), ("mlp", LINEARS_MLP)):
                for lin in lins:
                    p = f"model.layers.{i}.{blk}.{lin}"
                    d[f"{p}.weight"] = rnd(merged_weight(sd, p, cfg, ad), emulate)
        out[ad] = d
    return out
