"""Thin torch-tensor wrappers over the C ABI (include/mc_hip.h).  PyTorch provides device memory and
the current HIP stream; all arithmetic happens in libmc_hip.so.  No CPU fallback."""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import threading
from typing import Optional, Sequence

import torch

from . import _lib

ACT = {None: 0, "none": 0, "gelu": 1, "quick_gelu": 2, "silu": 3, "relu": 4}
# mc_gemm_args.family (include/mc_hip.h): which kernel family runs a linear.  Each family adds a row's products in ONE fp32 order whatever
# the number of rows; "auto" picks by M (strip for M <= 64, tile above), so a caller whose rows must not depend on how many other rows share
# the launch - the inference towers and projectors, whose row count is B x tokens - pins "tile" with `with ops.gemm_family("tile"):`.
FAMILY = {None: 0, "auto": 0, "strip": 1, "tile": 2}
_family_tls = threading.local()


def _fam(family=None) -> int:
    return FAMILY[family] if family is not None else getattr(_family_tls, "v", 0)


@contextlib.contextmanager
def gemm_family(name):
    """Default kernel family of every linear() / linear_ex() / linear_grouped() call of this thread inside the block."""
    old = getattr(_family_tls, "v", 0)
    _family_tls.v = FAMILY[name]
    try:
        yield
    finally:
        _family_tls.v = old

BF16 = _lib.storage_dtype()      # the library's 16-bit storage element: bf16, or fp16 with MC_STORAGE_DTYPE=fp16 (_lib.set_storage_dtype)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _req(t: torch.Tensor, dtype="storage", name="tensor"):
    if isinstance(dtype, str):
        dtype = BF16                                       # the library's storage element (rebound by _lib.set_storage_dtype)
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device (HIP) tensor; this path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class PackedWeight:
    """A linear weight in the MFMA-fragment layout of csrc/gemm.hip (+ optional bf16 bias)."""
    __slots__ = ("data", "N", "K", "Kp", "bias")

    def __init__(self, data, N, K, bias=None):
        self.data, self.N, self.K, self.Kp, self.bias = data, N, K, ceil_to(K, 64), bias

    @property
    def nbytes(self):
        return self.data.numel() * 2


def packed_elems(N: int, K: int) -> int:
    return ceil_to(N, 16) * ceil_to(K, 64)


def pack_weight(w: torch.Tensor, bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> PackedWeight:
    """[N, K] row-major (any float dtype, device) -> PackedWeight."""
    w = _req(w.to(BF16), BF16, "weight")
    if w.dim() != 2:
        raise ValueError("weight must be 2-D [N, K]")
    if w.stride(1) != 1:
        w = w.contiguous()
    N, K = w.shape
    if out is None:
        out = torch.empty(packed_elems(N, K), dtype=BF16, device=w.device)
    _lib.check(_lib.lib().mc_pack_weight_bf16(_p(w), w.stride(0), _p(out), N, K, _stream()), "mc_pack_weight_bf16")
    b = None if bias is None else _req(bias.to(BF16).contiguous(), BF16, "bias")
    return PackedWeight(out, N, K, b)


def unpack_weight(pw: PackedWeight) -> torch.Tensor:
    out = torch.empty(pw.N, pw.K, dtype=BF16, device=pw.data.device)
    _lib.check(_lib.lib().mc_unpack_weight_bf16(_p(pw.data), _p(out), pw.N, pw.K, _stream()), "mc_unpack_weight_bf16")
    return out


def compose_weight(w: Optional[torch.Tensor], terms: Sequence, N: int, K: int, rowmajor_out: bool = False):
    """W' = W + Σ scale·B·A  ->  PackedWeight.  terms = [(A [r,K], B [N,r], scale), ...] (device tensors)."""
    dev = (w if w is not None else terms[0][0]).device
    n = len(terms)
    r = 0
    ats, bs = [], []
    for (a, b, s) in terms:
        r0 = a.shape[0]
        rp = ceil_to(r0, 32)
        at = a.to(BF16).t().contiguous()                 # [K, r]
        bb = b.to(BF16).contiguous()                     # [N, r]
        if rp != r0:
            at = torch.nn.functional.pad(at, (0, rp - r0))
            bb = torch.nn.functional.pad(bb, (0, rp - r0))
        if r and rp != r:
            raise ValueError("all LoRA terms of one linear must share the rank")
        r = rp
        ats.append(at)
        bs.append(bb)
    out = torch.empty(packed_elems(N, K), dtype=BF16, device=dev)
    rm = torch.empty(N, K, dtype=BF16, device=dev) if rowmajor_out else None
    if w is not None:
        w = _req(w.to(BF16), BF16, "w")
        if w.stride(1) != 1:
            w = w.contiguous()
    at_arr = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in ats])
    b_arr = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in bs])
    sc = (C.c_float * max(n, 1))(*[float(t[2]) for t in terms])
    _lib.check(_lib.lib().mc_compose_weight_bf16(_p(w), 0 if w is None else w.stride(0), at_arr, b_arr, sc, n, r, _p(out),
                                                 _p(rm), K, N, K, _stream()), "mc_compose_weight_bf16")
    pw = PackedWeight(out, N, K)
    return (pw, rm) if rowmajor_out else pw


def linear(x: torch.Tensor, w: PackedWeight, act=None, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           out_f32: bool = False, alpha: float = 1.0, bias: bool = True, beta: float = 1.0, auto_split: bool = False, family=None) -> torch.Tensor:
    """out[M,N] = act(alpha * x W^T + b) + beta * residual.  x: [M, Kp] bf16 (Kp = K padded to 64, pad columns zero).
    auto_split: let the library split K when the launch would under-fill the GPU (training-time rank projections; the choice depends
    on M, so inference leaves it off to stay batch-invariant).  family: "auto" / "strip" / "tile" (default: the thread's gemm_family)."""
    _req(x, BF16, "x")
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be 2-D with contiguous rows")
    M, Kx = x.shape
    if Kx != w.Kp:
        raise ValueError(f"x has {Kx} columns, weight expects K padded to {w.Kp}")
    if out is None:
        out = torch.empty(M, w.N, dtype=torch.float32 if out_f32 else BF16, device=x.device)
    b = w.bias if bias else None
    a = _lib.GemmArgsC(x.data_ptr(), x.stride(0), w.data.data_ptr(), 0 if b is None else b.data_ptr(), 0 if residual is None else residual.data_ptr(),
                       0 if residual is None else residual.stride(0), out.data_ptr(), out.stride(0), M, w.N, w.Kp, ACT[act],
                       1 if out_f32 else 0, alpha, beta, 0, 0, -1 if auto_split else 1, 0.0, 0, 0, 0.0, 0 if auto_split else _fam(family))
    _lib.check(_lib.lib().mc_gemm_ex_bf16(C.byref(a), _stream()), "mc_gemm_ex_bf16")
    return out


def linear_ex(x: torch.Tensor, w: PackedWeight, row_scale: Optional[torch.Tensor] = None, swiglu: bool = False,
              residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, out_f32: bool = False, rms_eps: float = 0.0,
              rope=None, rms_out: Optional[torch.Tensor] = None, rms_out_eps: float = 0.0, family=None) -> torch.Tensor:
    """mc_gemm_ex_bf16: row_scale fp32 [M] (1/rms of a folded RMSNorm), swiglu (gate/up interleaved per 16 rows -> [M, N/2]),
    rms_eps > 0 (strip family): the RMSNorm factor computed inside the launch; rope = rope_scatter(...): the launch is a prefill's q|k|v
    projection, RoPE and the q / KV-cache scatter happen in its epilogue (the returned buffer is scratch then)."""
    _req(x, BF16, "x")
    M, Kx = x.shape
    if Kx != w.Kp:
        raise ValueError(f"x has {Kx} columns, weight expects K padded to {w.Kp}")
    n_out = w.N // 2 if swiglu else w.N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.float32 if out_f32 else BF16, device=x.device)
    a = _lib.GemmArgsC(x.data_ptr(), x.stride(0), w.data.data_ptr(), 0, 0 if residual is None else residual.data_ptr(),
                       0 if residual is None else residual.stride(0), out.data_ptr(), out.stride(-2), M, w.N, w.Kp, 0,
                       1 if out_f32 else 0, 1.0, 1.0, 0 if row_scale is None else row_scale.data_ptr(),
                       1 if swiglu else 0, 1, float(rms_eps), 0 if rope is None else C.addressof(rope[0]),
                       0 if rms_out is None else rms_out.data_ptr(), float(rms_out_eps), _fam(family))
    _lib.check(_lib.lib().mc_gemm_ex_bf16(C.byref(a), _stream()), "mc_gemm_ex_bf16")
    return out


def rope_scatter(row_b, row_pos, row_t, cos, sin, q_out, k_cache, v_cache, H, Hkv, D, Lq, Smax):
    """struct mc_rope_scatter for linear_ex / linear_grouped (returns (struct, tensors kept alive))."""
    r = _lib.RopeScatterC(*[t.data_ptr() for t in (row_b, row_pos, row_t, cos, sin, q_out, k_cache, v_cache)], H, Hkv, D, Lq, Smax)
    return (r, (row_b, row_pos, row_t, cos, sin, q_out, k_cache, v_cache))


def linear_grouped(x: torch.Tensor, weights: Sequence[PackedWeight], group_start: Sequence[int], row_scale: Optional[torch.Tensor] = None,
                   swiglu: bool = False, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, rope=None,
                   rms_out: Optional[torch.Tensor] = None, rms_out_eps: float = 0.0, family=None) -> torch.Tensor:
    """mc_gemm_grouped_bf16: rows [group_start[g], group_start[g+1]) of x use weights[g] (routed LocalLoRA linear)."""
    _req(x, BF16, "x")
    w0 = weights[0]
    M = x.shape[0]
    n_out = w0.N // 2 if swiglu else w0.N
    if out is None:
        out = torch.empty(M, n_out, dtype=BF16, device=x.device)
    a = _lib.GemmArgsC(x.data_ptr(), x.stride(0), 0, 0, 0 if residual is None else residual.data_ptr(),
                       0 if residual is None else residual.stride(0), out.data_ptr(), out.stride(0), 0, w0.N, w0.Kp, 0, 0, 1.0, 1.0,
                       0 if row_scale is None else row_scale.data_ptr(), 1 if swiglu else 0, 1, 0.0, 0 if rope is None else C.addressof(rope[0]),
                       0 if rms_out is None else rms_out.data_ptr(), float(rms_out_eps), _fam(family))
    gs = (C.c_int32 * len(group_start))(*group_start)
    wp = (C.c_void_p * len(weights))(*[w.data.data_ptr() for w in weights])
    _lib.check(_lib.lib().mc_gemm_grouped_bf16(C.byref(a), len(weights), gs, wp, _stream()), "mc_gemm_grouped_bf16")
    return out


def rms_scale(x, eps):
    _req(x, BF16, "x")
    M, D = x.shape
    rs = torch.empty(M, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mc_rms_scale_bf16(_p(x), x.stride(0), _p(rs), M, D, eps, _stream()), "mc_rms_scale_bf16")
    return rs


def rmsnorm(x, w, eps, out=None):
    _req(x, BF16, "x")
    M, D = x.shape
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().mc_rmsnorm_bf16(_p(x), x.stride(0), _p(w), _p(out), out.stride(0), M, D, eps, _stream()), "mc_rmsnorm_bf16")
    return out


def layernorm(x, w, b, eps, out=None):
    _req(x, BF16, "x")
    M, D = x.shape
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().mc_layernorm_bf16(_p(x), x.stride(0), _p(w), _p(b), _p(out), out.stride(0), M, D, eps, _stream()), "mc_layernorm_bf16")
    return out


def add_layernorm(x, table, idx, w, b, eps, sum_out=None, out=None):
    """(sum, LayerNorm(sum)) with sum[r] = bf16(x[r] + table[idx[r]]) in one pass (mc_add_layernorm_bf16); sum_out may alias x."""
    _req(x, BF16, "x")
    M, D = x.shape
    sum_out = torch.empty_like(x) if sum_out is None else sum_out
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().mc_add_layernorm_bf16(_p(x), x.stride(0), _p(table), table.stride(0), _p(idx), _p(sum_out), sum_out.stride(0), _p(w), _p(b),
                                                _p(out), out.stride(0), M, D, eps, _stream()), "mc_add_layernorm_bf16")
    return sum_out, out


def attn_mask(key_valid: Optional[torch.Tensor] = None, batch_split=None):
    """struct mc_attn_mask or None: key_valid uint8 [B, >= S] (0 = masked key), batch_split = (b_inner, inner_stride) - the two-level batch
    index of the Lq, S <= 8 kernel.  The returned struct holds raw pointers: keep key_valid alive until the launch has been issued."""
    if key_valid is None and batch_split is None:
        return None
    bi, st = (0, 0) if batch_split is None else (int(batch_split[0]), int(batch_split[1]))
    return _lib.AttnMaskC(0 if key_valid is None else key_valid.data_ptr(), 0 if key_valid is None else key_valid.stride(0), bi, st)


def _mask_ref(m):
    return None if m is None else C.byref(m)


def rope_kv(qkv, row_b, row_pos, row_t, cos, sin, q_out, k_cache, v_cache, H, Hkv, D, Lq, Smax):
    M = qkv.shape[0]
    _lib.check(_lib.lib().mc_rope_kv_bf16(_p(qkv), qkv.stride(0), _p(row_b), _p(row_pos), _p(row_t), _p(cos), _p(sin), _p(q_out),
                                          _p(k_cache), _p(v_cache), M, H, Hkv, D, Lq, Smax, _stream()), "mc_rope_kv_bf16")


def attn_prefill(q, k, v, out, B, H, Hkv, Lq, S, D, q_strides, k_strides, v_strides, o_row_stride, causal, q_offset=0,
                 scale=None, out_map=None, kv_lens=None, rel_table=None, rel_off=0, q_gate=None, key_valid=None, batch_split=None):
    scale = (1.0 / math.sqrt(D)) if scale is None else scale
    mk = attn_mask(key_valid, batch_split)
    _lib.check(_lib.lib().mc_attn_prefill_bf16(_p(q), *q_strides, _p(k), *k_strides, _p(v), *v_strides, _p(out), o_row_stride,
                                               _p(out_map), _p(kv_lens), B, H, Hkv, Lq, S, D, 1 if causal else 0, q_offset,
                                               scale, _p(rel_table), 0 if rel_table is None else rel_table.stride(0), rel_off,
                                               _p(q_gate), _mask_ref(mk), _stream()), "mc_attn_prefill_bf16")
    return out


def attn_probs(q, k, B, H, Hkv, Lq, S, D, q_strides, k_strides, causal=True, q_offset=0, scale=None, kv_lens=None, key_valid=None, out=None):
    """softmax(scale q k^T + mask) itself, (B, H, Lq, S) in the storage dtype (mc_attn_probs_bf16: output_attentions)."""
    scale = (1.0 / math.sqrt(D)) if scale is None else scale
    if out is None:
        out = torch.empty(B, H, Lq, S, dtype=q.dtype, device=q.device)
    _lib.check(_lib.lib().mc_attn_probs_bf16(_p(q), *q_strides, _p(k), *k_strides, _p(kv_lens), _p(key_valid),
                                             0 if key_valid is None else key_valid.stride(0), _p(out), B, H, Hkv, Lq, S, D,
                                             1 if causal else 0, q_offset, scale, _stream()), "mc_attn_probs_bf16")
    return out


def decode_workspace(B, H, D, S, device):
    """fp32 scratch of the decode attention: one partial per (b, h) and 512-key chunk of a cache of S positions (+ the new token's)."""
    n = C.c_int64(0)
    _lib.check(_lib.lib().mc_attn_decode_workspace_bytes(B, H, D, S, C.byref(n)), "mc_attn_decode_workspace_bytes")
    return torch.empty(n.value // 4, dtype=torch.float32, device=device)


def attn_decode(q, k, v, out, B, H, Hkv, S, D, q_strides, k_strides, v_strides, o_sb, nsplit=1, workspace=None, scale=None,
                kv_lens=None, key_valid=None):
    scale = (1.0 / math.sqrt(D)) if scale is None else scale
    if workspace is None and (nsplit > 1 or S > 19968):
        workspace = decode_workspace(B, H, D, S, q.device)
    _lib.check(_lib.lib().mc_attn_decode_bf16(_p(q), *q_strides, _p(k), *k_strides, _p(v), *v_strides, _p(out), o_sb,
                                              _p(workspace), _p(kv_lens), B, H, Hkv, S, D, nsplit, scale, _mask_ref(attn_mask(key_valid)), _stream()),
               "mc_attn_decode_bf16")
    return out


def silu_mul(gate_up, I, out=None):
    M = gate_up.shape[0]
    out = torch.empty(M, I, dtype=BF16, device=gate_up.device) if out is None else out
    _lib.check(_lib.lib().mc_silu_mul_bf16(_p(gate_up), gate_up.stride(0), _p(out), out.stride(0), M, I, _stream()), "mc_silu_mul_bf16")
    return out


def copy_rows(src, dst, n_rows, src_idx=None, dst_idx=None):
    D = src.shape[-1]
    _lib.check(_lib.lib().mc_copy_rows_bf16(_p(src), src.stride(-2), _p(src_idx), _p(dst), dst.stride(-2), _p(dst_idx), n_rows, D,
                                            _stream()), "mc_copy_rows_bf16")
    return dst


def embed_rows(table, ids, dst, dst_idx=None):
    n = ids.numel()
    _lib.check(_lib.lib().mc_embed_rows_bf16(_p(table), table.stride(0), _p(ids), _p(dst), dst.stride(-2), _p(dst_idx), n,
                                             table.shape[1], _stream()), "mc_embed_rows_bf16")
    return dst


def argmax(logits_f32, out=None):
    M, N = logits_f32.shape
    out = torch.empty(M, dtype=torch.int64, device=logits_f32.device) if out is None else out
    _lib.check(_lib.lib().mc_argmax_f32(_p(logits_f32), logits_f32.stride(0), _p(out), M, N, _stream()), "mc_argmax_f32")
    return out


def log_softmax(logits_f32, out=None):
    """mc_log_softmax_f32: rows of fp32 logits -> log-probabilities (beam search scoring)."""
    M, N = logits_f32.shape
    out = torch.empty(M, N, dtype=torch.float32, device=logits_f32.device) if out is None else out
    _lib.check(_lib.lib().mc_log_softmax_f32(_p(logits_f32), logits_f32.stride(0), _p(out), out.stride(0), M, N, _stream()), "mc_log_softmax_f32")
    return out


def sample_step(logits_f32, temperature=1.0, top_k=0, top_p=1.0, seed=0, step=0, uniform=None, want_probs=False):
    """One sampled token per row of logits [M, V] fp32 (HF warper order: temperature, top-k, top-p, multinomial).  Returns ids int64 [M]
    (and the final probabilities [M, V] when want_probs)."""
    M, V = logits_f32.shape
    ids = torch.empty(M, dtype=torch.int64, device=logits_f32.device)
    probs = torch.empty(M, V, dtype=torch.float32, device=logits_f32.device) if want_probs else None
    _lib.check(_lib.lib().mc_sample_step_f32(_p(logits_f32), logits_f32.stride(0), _p(ids), None, 0, None, int(step), None, int(seed) & (2 ** 64 - 1),
                                             M, V, float(temperature), int(top_k), float(top_p), _p(uniform), _p(probs), V, _stream()),
               "mc_sample_step_f32")
    return (ids, probs) if want_probs else ids


def im2col(x, kh, kw, sh, sw, Kp=None):
    _req(x, BF16, "x")
    x = x.contiguous()
    B, Cc, Hh, Ww = x.shape
    K = Cc * kh * kw
    Kp = ceil_to(K, 64) if Kp is None else Kp
    oh, ow = (Hh - kh) // sh + 1, (Ww - kw) // sw + 1
    out = torch.empty(B * oh * ow, Kp, dtype=BF16, device=x.device)
    _lib.check(_lib.lib().mc_im2col_bf16(_p(x), _p(out), B, Cc, Hh, Ww, kh, kw, sh, sw, Kp, _stream()), "mc_im2col_bf16")
    return out, oh, ow


def vit_assemble(patches, cls, pos, B, T, D):
    out = torch.empty(B, T + (1 if cls is not None else 0), D, dtype=BF16, device=patches.device)
    _lib.check(_lib.lib().mc_vit_assemble_bf16(_p(patches), _p(cls), _p(pos), _p(out), B, T, D, _stream()), "mc_vit_assemble_bf16")
    return out


def add(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    _lib.check(_lib.lib().mc_add_bf16(_p(a), _p(b), _p(out), a.numel(), _stream()), "mc_add_bf16")
    return out


# ---- encoder-specific kernels ---------------------------------------------------------------------------
def add_rows(x, table, idx=None, out=None):
    """out[r] = x[r] + table[idx[r]] (idx None -> r)."""
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().mc_add_rows_bf16(_p(x), x.stride(0), _p(table), table.stride(0), _p(idx), _p(out), out.stride(0), x.shape[0],
                                           x.shape[1], _stream()), "mc_add_rows_bf16")
    return out


def zero_rows(x, rows):
    if rows.numel():
        _lib.check(_lib.lib().mc_zero_rows_bf16(_p(x), x.stride(0), _p(rows), rows.numel(), x.shape[1], _stream()), "mc_zero_rows_bf16")
    return x


def beats_padding(frame_mask_u8, B, T, span, x, kv_lens, bad_flag):
    """Device-side forward_padding_mask + x[padding_mask] = 0 of BEATs (mc_beats_padding_bf16): fills kv_lens [B] int32, zeroes the padded
    token rows of x [B * T, D], ORs 1 into bad_flag[0] when some clip's padding is not a suffix."""
    _lib.check(_lib.lib().mc_beats_padding_bf16(_p(frame_mask_u8), frame_mask_u8.stride(0), B, T, span, _p(x), x.stride(0), x.shape[1], _p(kv_lens),
                                                _p(bad_flag), _stream()), "mc_beats_padding_bf16")


def im2col_ex(x, strides, B, Cc, Hin, Win, c0, Cg, kh, kw, sh, sw, ph, pw, oh, ow, Kp=None):
    """x: bf16 device tensor addressed as x[b*s_b + c*s_c + y*s_h + x*s_w]."""
    K = Cg * kh * kw
    Kp = ceil_to(K, 64) if Kp is None else Kp
    out = torch.empty(B * oh * ow, Kp, dtype=BF16, device=x.device)
    _lib.check(_lib.lib().mc_im2col_ex_bf16(_p(x), *strides, _p(out), B, Cc, Hin, Win, c0, Cg, kh, kw, sh, sw, ph, pw, oh, ow, Kp,
                                            _stream()), "mc_im2col_ex_bf16")
    return out


def beats_gate(g8, grep_a, B, L, H):
    gate = torch.empty(B, H, L, dtype=torch.float32, device=g8.device)
    _lib.check(_lib.lib().mc_beats_gate_f32(_p(g8), _p(grep_a), _p(gate), B, L, H, _stream()), "mc_beats_gate_f32")
    return gate


def group_max(x, G, n, out=None, bcast=None):
    C_ = x.shape[1]
    if out is None and bcast is None:
        out = torch.empty(G, C_, dtype=BF16, device=x.device)
    _lib.check(_lib.lib().mc_group_max_bf16(_p(x), x.stride(0), _p(out), 0 if out is None else out.stride(0), _p(bcast),
                                            0 if bcast is None else bcast.stride(0), G, n, C_, _stream()), "mc_group_max_bf16")
    return out


def fps(pts, npoint, start_idx=None):
    B, N, Cc = pts.shape
    idx = torch.empty(B, npoint, dtype=torch.int32, device=pts.device)
    centers = torch.empty(B, npoint, 3, dtype=torch.float32, device=pts.device)
    _lib.check(_lib.lib().mc_fps_bf16(_p(pts), B, N, Cc, _p(start_idx), npoint, _p(idx), _p(centers), _stream()), "mc_fps_bf16")
    return idx, centers


def knn_group(pts, centers, k, Kp=64):
    B, N, Cc = pts.shape
    G = centers.shape[1]
    out = torch.empty(B * G * k, Kp, dtype=BF16, device=pts.device)
    idx = torch.empty(B, G, k, dtype=torch.int32, device=pts.device)
    _lib.check(_lib.lib().mc_knn_group_bf16(_p(pts), B, N, Cc, _p(centers), G, k, _p(out), Kp, _p(idx), _stream()), "mc_knn_group_bf16")
    return out, idx


def f32_rows_to_bf16(x, Kp=64):
    rows, Cc = x.shape
    out = torch.empty(rows, Kp, dtype=BF16, device=x.device)
    _lib.check(_lib.lib().mc_f32_rows_to_bf16(_p(x), Cc, _p(out), Kp, rows, _stream()), "mc_f32_rows_to_bf16")
    return out


# ---- training step (csrc/train.hip, csrc/attention_bwd.hip) ---------------------------------------------
def transpose(x, Rp=None, out=None):
    """[R, C] bf16 (row stride free) -> [C, Rp] with zero padded columns R..Rp-1."""
    R, Cc = x.shape
    Rp = R if Rp is None else Rp
    out = torch.empty(Cc, Rp, dtype=BF16, device=x.device) if out is None else out
    _lib.check(_lib.lib().mc_transpose_bf16(_p(x), x.stride(0), _p(out), out.stride(0), R, Cc, Rp, _stream()), "mc_transpose_bf16")
    return out


_tn_ws = {}


def gemm_tn(a_list, b_list, out_list, alpha=1.0):
    """out_i[P, Q] (fp32) = alpha * a_i^T b_i for up to 3 same-shape problems; a_i [M, P], b_i [M, Q] bf16 with contiguous rows."""
    a0, b0 = a_list[0], b_list[0]
    M, P = a0.shape
    Q = b0.shape[1]
    n = len(a_list)
    L = _lib.lib()
    need = C.c_int64(0)
    _lib.check(L.mc_gemm_tn_workspace_floats(M, P, Q, n, C.byref(need)), "mc_gemm_tn_workspace_floats")
    ws = None
    if need.value:
        ws = _tn_ws.get(a0.device)
        if ws is None or ws.numel() < need.value:
            ws = torch.empty(need.value, dtype=torch.float32, device=a0.device)
            _tn_ws[a0.device] = ws
    for t in list(a_list) + list(b_list):
        if t.stride(1) != 1 or t.dtype != BF16:
            raise ValueError("gemm_tn operands must be bf16 with contiguous rows")
    if any(t.stride(0) != a0.stride(0) for t in a_list) or any(t.stride(0) != b0.stride(0) for t in b_list) or \
            any(o.stride(0) != out_list[0].stride(0) or o.dtype != torch.float32 for o in out_list):
        raise ValueError("gemm_tn: the problems of one launch must share strides; outputs are fp32")
    pa = (C.c_void_p * n)(*[t.data_ptr() for t in a_list])
    pb = (C.c_void_p * n)(*[t.data_ptr() for t in b_list])
    po = (C.c_void_p * n)(*[t.data_ptr() for t in out_list])
    _lib.check(L.mc_gemm_tn_bf16(pa, a0.stride(0), pb, b0.stride(0), po, out_list[0].stride(0), n, M, P, Q, float(alpha), _p(ws), _stream()),
               "mc_gemm_tn_bf16")
    return out_list


def pack_weight_t(w_t: torch.Tensor, out: Optional[torch.Tensor] = None) -> PackedWeight:
    """Packed form of W = w_t^T for a row-major w_t [K, N] (no transposed copy in HBM)."""
    _req(w_t, BF16, "w_t")
    K, N = w_t.shape
    if w_t.stride(1) != 1:
        raise ValueError("w_t must have contiguous rows")
    data = torch.empty(packed_elems(N, K), dtype=BF16, device=w_t.device) if out is None else out
    _lib.check(_lib.lib().mc_pack_weight_strided_bf16(_p(w_t), 1, w_t.stride(0), _p(data), N, K, _stream()), "mc_pack_weight_strided_bf16")
    return PackedWeight(data, N, K)


def lora_mask_rows(t, row_adapter, r, n_adapters):
    """t [M, n_linears * n_adapters * r]: zero every r-wide block that does not belong to the row's adapter."""
    _lib.check(_lib.lib().mc_lora_mask_rows_bf16(_p(t), t.stride(0), _p(row_adapter), t.shape[0], r, n_adapters, t.shape[1], _stream()),
               "mc_lora_mask_rows_bf16")
    return t


def rmsnorm_bwd(x, g, dy, eps, dres=None, out=None):
    M, D = x.shape
    out = torch.empty(M, D, dtype=BF16, device=x.device) if out is None else out
    _lib.check(_lib.lib().mc_rmsnorm_bwd_bf16(_p(x), x.stride(0), _p(g), _p(dy), dy.stride(0), _p(dres), 0 if dres is None else dres.stride(0),
                                              _p(out), out.stride(0), M, D, eps, _stream()), "mc_rmsnorm_bwd_bf16")
    return out


def swiglu_bwd(gu, dinter, out=None):
    M, I2 = gu.shape
    out = torch.empty(M, I2, dtype=BF16, device=gu.device) if out is None else out
    _lib.check(_lib.lib().mc_swiglu_bwd_bf16(_p(gu), gu.stride(0), _p(dinter), dinter.stride(0), _p(out), out.stride(0), M, I2 // 2, _stream()),
               "mc_swiglu_bwd_bf16")
    return out


def act(pre, kind, dy=None):
    """dy None: act(pre); else dy * act'(pre).  Contiguous tensors."""
    out = torch.empty_like(pre)
    _lib.check(_lib.lib().mc_act_bf16(_p(pre), _p(dy), _p(out), pre.numel(), ACT[kind], _stream()), "mc_act_bf16")
    return out


def ce_loss(logits_f32, labels, inv_n, want_grad=True):
    """labels int64 [M] (already shifted, < 0 ignored) -> (loss_rows fp32 [M], dlogits bf16 [M, V] = (softmax - onehot) * inv_n);
    want_grad=False skips the gradient (dlogits is None)."""
    M, V = logits_f32.shape
    loss_rows = torch.empty(M, dtype=torch.float32, device=logits_f32.device)
    dl = torch.empty(M, V, dtype=BF16, device=logits_f32.device) if want_grad else None
    _lib.check(_lib.lib().mc_ce_loss_f32(_p(logits_f32), logits_f32.stride(0), _p(labels), _p(loss_rows), _p(dl), dl.stride(0) if want_grad else 0,
                                         M, V, inv_n, _stream()), "mc_ce_loss_f32")
    return loss_rows, dl


def colsum(x, out=None):
    M, Cc = x.shape
    out = torch.empty(Cc, dtype=torch.float32, device=x.device) if out is None else out
    _lib.check(_lib.lib().mc_colsum_bf16(_p(x), x.stride(0), _p(out), M, Cc, _stream()), "mc_colsum_bf16")
    return out


def rope_inplace(x, row_pos, cos, sin, n_heads, D, sign):
    _lib.check(_lib.lib().mc_rope_inplace_bf16(_p(x), x.stride(0), _p(row_pos), _p(cos), _p(sin), x.shape[0], n_heads, D, sign, _stream()),
               "mc_rope_inplace_bf16")
    return x


def attn_prefill_lse(q, k, v, out, lse, B, H, Lq, S, D, q_strides, k_strides, v_strides, o_row_stride, causal, scale=None, kv_lens=None,
                     dropout=None):
    """dropout = (p, seed, stream_id): dropout on the attention probabilities (mc_attn_prefill_dropout_bf16); pass the same triple to attn_bwd."""
    sc = (1.0 / math.sqrt(D)) if scale is None else scale
    if dropout is not None and dropout[0] > 0:
        _lib.check(_lib.lib().mc_attn_prefill_dropout_bf16(_p(q), *q_strides, _p(k), *k_strides, _p(v), *v_strides, _p(out), o_row_stride,
                                                           _p(kv_lens), B, H, H, Lq, S, D, 1 if causal else 0, 0, sc, _p(lse), float(dropout[0]),
                                                           int(dropout[1]) & (2 ** 64 - 1), int(dropout[2]), None, _stream()), "mc_attn_prefill_dropout_bf16")
        return out
    _lib.check(_lib.lib().mc_attn_prefill_lse_bf16(_p(q), *q_strides, _p(k), *k_strides, _p(v), *v_strides, _p(out), o_row_stride, None,
                                                   _p(kv_lens), B, H, H, Lq, S, D, 1 if causal else 0, 0, sc, None, 0, 0, None, _p(lse),
                                                   None, _stream()), "mc_attn_prefill_lse_bf16")
    return out


def attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, H, Lq, S, D, q_strides, k_strides, v_strides, o_strides, dq_strides, dk_strides, dv_strides,
             causal, scale=None, kv_lens=None, dropout=None):
    delta = torch.empty(B * H * Lq, dtype=torch.float32, device=q.device)
    dp, dseed, dstream = (float(dropout[0]), int(dropout[1]) & (2 ** 64 - 1), int(dropout[2])) if dropout is not None else (0.0, 0, 0)
    a = _lib.AttnBwdArgsC(q.data_ptr(), *q_strides, k.data_ptr(), *k_strides, v.data_ptr(), *v_strides, o.data_ptr(), d_o.data_ptr(), *o_strides,
                          lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), *dq_strides, dk.data_ptr(), *dk_strides, dv.data_ptr(), *dv_strides,
                          0 if kv_lens is None else kv_lens.data_ptr(), B, H, Lq, S, D, 1 if causal else 0, 0,
                          (1.0 / math.sqrt(D)) if scale is None else scale, dp, dseed, dstream)
    _lib.check(_lib.lib().mc_attn_bwd_bf16(C.byref(a), _stream()), "mc_attn_bwd_bf16")


def adamw(p32, g32, m, v, p16, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    _lib.check(_lib.lib().mc_adamw_f32(_p(p32), _p(g32), _p(m), _p(v), _p(p16), p32.numel(), lr, beta1, beta2, eps, wd, step, grad_scale, _stream()),
               "mc_adamw_f32")


def adamw_segments(p32, g32, m, v, p16, segs_dev, n_segs, lr, lr_alt, beta1, beta2, eps, wd, step, grad_scale=1.0):
    """AdamW with two learning-rate groups chosen per element (mc_adamw_segments_f32; segs_dev: uint8 tensor of mc_adamw_seg records)."""
    _lib.check(_lib.lib().mc_adamw_segments_f32(_p(p32), _p(g32), _p(m), _p(v), _p(p16), _p(segs_dev), n_segs, lr, lr_alt, beta1, beta2, eps, wd,
                                                step, grad_scale, _stream()), "mc_adamw_segments_f32")


def axpy_f32(y, x, alpha=1.0):
    """y += alpha * x (flat fp32)."""
    assert y.dtype == torch.float32 and x.dtype == torch.float32 and y.numel() == x.numel()
    _lib.check(_lib.lib().mc_axpy_f32(_p(y), _p(x), y.numel(), float(alpha), _stream()), "mc_axpy_f32")


def layernorm_bwd(x, g, dy, eps, want_t=True):
    """-> (dx, t = dy * xhat or None): LayerNorm backward w.r.t. the input; dgamma = colsum(t), dbeta = colsum(dy)."""
    M, D = x.shape
    dx = torch.empty(M, D, dtype=BF16, device=x.device)
    t = torch.empty(M, D, dtype=BF16, device=x.device) if want_t else None
    _lib.check(_lib.lib().mc_layernorm_bwd_bf16(_p(x), x.stride(0), _p(g), _p(dy), dy.stride(0), _p(dx), dx.stride(0), _p(t),
                                                0 if t is None else t.stride(0), M, D, float(eps), _stream()), "mc_layernorm_bwd_bf16")
    return dx, t


def dropout(x, p: float, seed: int, stream_id: int, out=None, accumulate: bool = False, alpha: float = 1.0):
    """out = (accumulate ? out : 0) + alpha * x * keep / (1 - p); keep from Philox4x32-10 keyed by (seed, stream_id, element index)."""
    M, K = x.shape
    out = torch.empty(M, K, dtype=BF16, device=x.device) if out is None else out
    _lib.check(_lib.lib().mc_dropout_bf16(_p(x), x.stride(0), _p(out), out.stride(0), M, K, float(p), int(seed) & (2 ** 64 - 1), int(stream_id),
                                          int(accumulate), float(alpha), _stream()), "mc_dropout_bf16")
    return out


def cast_bf16(x32, out=None):
    out = torch.empty(x32.shape, dtype=BF16, device=x32.device) if out is None else out
    _lib.check(_lib.lib().mc_cast_f32_bf16(_p(x32), _p(out), x32.numel(), _stream()), "mc_cast_f32_bf16")
    return out


def attn_decode_rope(qkv, cos, sin, k_cache, v_cache, out, kv_lens, B, H, Hkv, Smax, D, nsplit=1, workspace=None, scale=None, key_valid=None):
    """Decode attention with RoPE + KV append fused (caches [B, Hkv, Smax, D]); kv_lens counts the token being decoded.
    qkv: the [B, (H + 2 Hkv) D] bf16 rows.  nsplit is a launch shape, not part of the result."""
    sc = (1.0 / math.sqrt(D)) if scale is None else scale
    st = (Hkv * Smax * D, D, Smax * D)
    if workspace is None and (nsplit > 1 or Smax > 19968):
        workspace = decode_workspace(B, H, D, Smax, out.device)
    _lib.check(_lib.lib().mc_attn_decode_rope_bf16(_p(qkv), qkv.stride(0), _p(cos), _p(sin), _p(k_cache), *st, _p(v_cache), *st, _p(out),
                                                   out.stride(0), _p(workspace), _p(kv_lens), B, H, Hkv, Smax, D, nsplit, sc,
                                                   _mask_ref(attn_mask(key_valid)), _stream()),
               "mc_attn_decode_rope_bf16")
    return out
