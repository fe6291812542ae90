"""ctypes binding of libmc_hip.so (C ABI: include/mc_hip.h).  Fails loudly if the library is missing."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmc_hip.so")
# The same sources instantiated on IEEE-half storage (csrc/common.h, -DMC_STORAGE_F16): the reference's own inference dtype
# (modelcompose/model/builder.py:41, :162, :185), kept as the parity instrument.  One storage dtype per process: MC_STORAGE_DTYPE=fp16 in
# the environment, or set_storage_dtype("fp16") before anything is built.
LIB_PATHS = {"bf16": LIB_PATH, "fp16": os.path.join(_HERE, "libmc_hip_f16.so")}
if os.environ.get("MC_PROBES_LIB", "0") == "1":      # the probes build (bf16 only): tools/probes/*.py
    LIB_PATHS["bf16"] = LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "probes", "libmc_hip_probes.so")
ABI_VERSION = 10
_DTYPE_CODES = {"bf16": 1, "fp16": 2}                        # MC_DTYPE_BF16 / MC_DTYPE_F16 of mc_hip.h


def _norm_dtype(name) -> str:
    n = str(name).replace("torch.", "").lower()
    if n in ("bf16", "bfloat16"):
        return "bf16"
    if n in ("fp16", "f16", "float16", "half"):
        return "fp16"
    raise ValueError(f"storage dtype must be bf16 or fp16, not {name!r}")


_storage = _norm_dtype(os.environ.get("MC_STORAGE_DTYPE", "bf16"))
_libs: dict = {}
_lib = None


def storage_name() -> str:
    return _storage


def storage_dtype():
    """torch dtype of the 16-bit storage element of the library this process uses."""
    import torch
    return torch.float16 if _storage == "fp16" else torch.bfloat16


def set_storage_dtype(name) -> str:
    """Select the library instantiation for this process: "bf16" (default, the headline) or "fp16".  One storage dtype per process: the
    two libraries have their own global state (stream workspaces, options) and tensors built for one must never reach the other, so the
    switch is only allowed BEFORE the first library has been loaded - i.e. before any model, packed weight or handle exists.
    Normally the choice is made by MC_STORAGE_DTYPE in the environment (bench.py --dtype, the fp16 tests' subprocesses)."""
    global _storage, _lib
    import sys
    new = _norm_dtype(name)
    if new != _storage:
        if _libs:
            raise MCError(f"storage dtype is {_storage} for this process (a library is already loaded): set MC_STORAGE_DTYPE={new} before the "
                          f"first use instead of switching")
        _storage = new
        _lib = None
    dt = storage_dtype()
    for mname, mod in list(sys.modules.items()):
        if mname.startswith("modelcompose_amd") and mod is not None and hasattr(mod, "BF16"):
            mod.BF16 = dt
    return _storage


c_p = C.c_void_p
c_i = C.c_int
c_l = C.c_int64
c_f = C.c_float

# name -> argtypes  (every function returns int status unless noted)
_SIGS = {
    "mc_abi_version": [],
    "mc_storage_dtype": [],
    "mc_device_info": [C.POINTER(c_i), C.POINTER(c_l), C.c_char_p, c_i],
    "mc_packed_weight_elems": [c_i, c_i, C.POINTER(c_l)],
    "mc_pack_weight_bf16": [c_p, c_l, c_p, c_i, c_i, c_p],
    "mc_unpack_weight_bf16": [c_p, c_p, c_i, c_i, c_p],
    "mc_compose_weight_bf16": [c_p, c_l, C.POINTER(c_p), C.POINTER(c_p), C.POINTER(c_f), c_i, c_i, c_p, c_p, c_l, c_i, c_i, c_p],
    "mc_gemm_bf16": [c_p, c_l, c_p, c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_p],
    "mc_compose_weight_ex_bf16": [c_p, c_l, C.POINTER(c_p), C.POINTER(c_p), C.POINTER(c_f), c_i, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_i, c_i, c_p, c_p],
    "mc_compose_weight_dither_bf16": [c_p, c_l, C.POINTER(c_p), C.POINTER(c_p), C.POINTER(c_f), c_i, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_i, c_i, c_p,
                                      C.c_uint32, c_p],
    "mc_rms_scale_bf16": [c_p, c_l, c_p, c_i, c_i, c_f, c_p],
    "mc_gemm_profile_enable": [c_i],
    "mc_gemm_profile_read": [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_l)],
    "mc_gemm_profile_read_bytes": [C.POINTER(C.c_double)],
    "mc_gemm_profile_read_range": [c_i, c_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_l)],
    "mc_gemm_set_option": [C.c_char_p, c_i],
    "mc_gemm_tn_workspace_floats": [c_i, c_i, c_i, c_i, C.POINTER(C.c_int64)],
    "mc_gemm_tn_bf16": [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_p, c_p],
    "mc_pack_weight_strided_bf16": [c_p, c_l, c_l, c_p, c_i, c_i, c_p],
    "mc_pack_weight_batch_bf16": [c_p, c_i, c_i, c_p],
    "mc_rmsnorm_bf16": [c_p, c_l, c_p, c_p, c_l, c_i, c_i, c_f, c_p],
    "mc_layernorm_bf16": [c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_i, c_f, c_p],
    "mc_add_layernorm_bf16": [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_i, c_f, c_p],
    "mc_rope_kv_bf16": [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "mc_attn_prefill_bf16": [c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_p, c_p,
                             c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_i, c_i, c_p, c_p, c_p],
    "mc_attn_decode_workspace_bytes": [c_i, c_i, c_i, c_i, C.POINTER(c_l)],
    "mc_attn_decode_bf16": [c_p, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_p, c_p,
                            c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p],
    "mc_gemm_reserve_workspace": [C.c_void_p],
    "mc_gemm_release_workspace": [C.c_void_p],
    "mc_silu_mul_bf16": [c_p, c_l, c_p, c_l, c_i, c_i, c_p],
    "mc_gather_last_rows_bf16": [C.c_void_p, C.c_int64, C.c_void_p, c_i, C.c_void_p, C.c_int64, c_i, c_i, C.c_void_p],
    "mc_copy_rows_bf16": [c_p, c_l, c_p, c_p, c_l, c_p, c_i, c_i, c_p],
    "mc_embed_rows_bf16": [c_p, c_l, c_p, c_p, c_l, c_p, c_i, c_i, c_p],
    "mc_argmax_f32": [c_p, c_l, c_p, c_i, c_i, c_p],
    "mc_im2col_bf16": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p],
    "mc_vit_assemble_bf16": [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "mc_add_bf16": [c_p, c_p, c_p, c_l, c_p],
}


class GemmArgsC(C.Structure):
    """struct mc_gemm_args (include/mc_hip.h)."""
    _fields_ = [("x", c_p), ("ldx", c_l), ("w_packed", c_p), ("bias", c_p), ("residual", c_p), ("ldr", c_l), ("out", c_p), ("ldo", c_l),
                ("M", c_i), ("N", c_i), ("K", c_i), ("act", c_i), ("out_f32", c_i), ("alpha", c_f), ("beta", c_f),
                ("row_scale", c_p), ("swiglu", c_i), ("split_k", c_i), ("rms_eps", c_f), ("rope", c_p), ("rms_out", c_p), ("rms_out_eps", c_f),
                ("family", c_i)]


class ComposeMultiArgsC(C.Structure):
    """struct mc_compose_multi_args (include/mc_hip.h)."""
    _fields_ = [("w", c_p), ("ldw", c_l), ("at_list", C.POINTER(c_p)), ("b_list", C.POINTER(c_p)), ("scales", C.POINTER(c_f)), ("n_terms", c_i), ("r", c_i),
                ("n_out", c_i), ("out_packed", C.POINTER(c_p)), ("out_rowmajor", C.POINTER(c_p)), ("term_mask", C.POINTER(C.c_uint32)),
                ("dither_seeds", C.POINTER(C.c_uint32)), ("retention_parts", C.POINTER(c_p)), ("ldo", c_l), ("N", c_i), ("K", c_i), ("col_scale", c_p),
                ("nb_stride", c_i), ("nb_offset", c_i)]


class AttnMaskC(C.Structure):
    """struct mc_attn_mask (include/mc_hip.h)."""
    _fields_ = [("key_valid", c_p), ("key_valid_stride", c_l), ("b_inner", c_i), ("inner_stride", c_l)]


class RopeScatterC(C.Structure):
    """struct mc_rope_scatter (include/mc_hip.h)."""
    _fields_ = [("row_b", c_p), ("row_pos", c_p), ("row_t", c_p), ("cos_table", c_p), ("sin_table", c_p), ("q_out", c_p), ("k_cache", c_p),
                ("v_cache", c_p), ("H", c_i), ("Hkv", c_i), ("D", c_i), ("Lq", c_i), ("Smax", c_i)]


class AttnBwdArgsC(C.Structure):
    """struct mc_attn_bwd_args (include/mc_hip.h)."""
    _fields_ = ([("q", c_p), ("q_sb", c_l), ("q_st", c_l), ("q_sh", c_l), ("k", c_p), ("k_sb", c_l), ("k_st", c_l), ("k_sh", c_l),
                 ("v", c_p), ("v_sb", c_l), ("v_st", c_l), ("v_sh", c_l), ("o", c_p), ("d_o", c_p), ("o_sb", c_l), ("o_st", c_l), ("o_sh", c_l),
                 ("lse", c_p), ("delta", c_p), ("dq", c_p), ("dq_sb", c_l), ("dq_st", c_l), ("dq_sh", c_l),
                 ("dk", c_p), ("dk_sb", c_l), ("dk_st", c_l), ("dk_sh", c_l), ("dv", c_p), ("dv_sb", c_l), ("dv_st", c_l), ("dv_sh", c_l),
                 ("kv_lens", c_p)] + [(n, c_i) for n in ("B", "H", "Lq", "S", "D", "causal", "q_offset")] + [("scale", c_f)] +
                [("dropout_p", c_f), ("dropout_seed", C.c_uint64), ("dropout_stream", C.c_uint32)])


class LlmConfigC(C.Structure):
    """struct mc_llm_config (include/mc_hip.h)."""
    _fields_ = [(n, C.c_int) for n in ("hidden", "inter", "n_layers", "n_heads", "n_kv_heads", "head_dim", "vocab", "n_adapters",
                                       "max_pos")] + [("rms_eps", C.c_float)]


_SIGS.update({
    "mc_gemm_ex_bf16": [C.POINTER(GemmArgsC), c_p],
    "mc_fbank_f32": [c_p, c_p, c_l, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_p, c_p, c_i, c_p],
    "mc_image_preprocess_u8": [c_p, c_i, c_i, c_i, c_i, c_i, c_i, C.POINTER(C.c_int32), c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                               C.POINTER(c_f), C.POINTER(c_f), c_p, c_p, c_p, c_p, c_p],
    "mc_video_preprocess_u8": [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, C.POINTER(c_f), C.POINTER(c_f), c_p, c_p, c_p],
    "mc_ties_hist": [c_p, c_i, c_l, c_l, c_i, c_i, c_i, c_p, c_i, c_p, c_p],
    "mc_ties_merge": [c_p, c_i, c_l, c_l, c_i, c_p, c_p, c_p, c_i, c_p, c_p],
    "mc_merge_metrics": [c_p, c_i, c_l, c_l, c_i, c_p, c_p, c_p],
    "mc_merge_metrics_blocks": [],
    "mc_gemm_grouped_bf16": [C.POINTER(GemmArgsC), c_i, C.POINTER(C.c_int32), C.POINTER(c_p), c_p],
    "mc_attn_bwd_bf16": [C.POINTER(AttnBwdArgsC), c_p],
    "mc_attn_decode_rope_bf16": [c_p, c_l, c_p, c_p, c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_p, c_p,
                                 c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p],
    "mc_attn_prefill_dropout_bf16": [c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_p,
                                     c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_f, C.c_uint64, C.c_uint32, c_p, c_p],
    "mc_attn_prefill_lse_bf16": [c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_p, c_p,
                                 c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_i, c_i, c_p, c_p, c_p, c_p],
    "mc_transpose_bf16": [c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p],
    "mc_lora_mask_rows_bf16": [c_p, c_l, c_p, c_i, c_i, c_i, c_i, c_p],
    "mc_rmsnorm_bwd_bf16": [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_f, c_p],
    "mc_swiglu_bwd_bf16": [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_p],
    "mc_act_bf16": [c_p, c_p, c_p, c_l, c_i, c_p],
    "mc_ce_loss_f32": [c_p, c_l, c_p, c_p, c_p, c_l, c_i, c_i, c_f, c_p],
    "mc_colsum_bf16": [c_p, c_l, c_p, c_i, c_i, c_p],
    "mc_rope_inplace_bf16": [c_p, c_l, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p],
    "mc_adamw_f32": [c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_p],
    "mc_adamw_segments_f32": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_p],
    "mc_axpy_f32": [c_p, c_p, c_l, c_f, c_p],
    "mc_cast_f32_bf16": [c_p, c_p, c_l, c_p],
    "mc_layernorm_bwd_bf16": [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_f, c_p],
    "mc_dropout_bf16": [c_p, c_l, c_p, c_l, c_i, c_i, c_f, C.c_uint64, C.c_uint32, c_i, c_f, c_p],
    "mc_add_rows_bf16": [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_i, c_i, c_p],
    "mc_zero_rows_bf16": [c_p, c_l, c_p, c_i, c_i, c_p],
    "mc_beats_padding_bf16": [c_p, c_l, c_i, c_i, c_i, c_p, c_l, c_i, c_p, c_p, c_p],
    "mc_im2col_ex_bf16": [c_p, c_l, c_l, c_l, c_l, c_p] + [c_i] * 15 + [c_p],
    "mc_beats_gate_f32": [c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "mc_group_max_bf16": [c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_p],
    "mc_fps_bf16": [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p],
    "mc_knn_group_bf16": [c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_p],
    "mc_f32_rows_to_bf16": [c_p, c_i, c_p, c_i, c_l, c_p],
    "mc_decode_state_init": [c_p, c_p, c_i, c_i, c_p],
    "mc_decode_state_advance": [c_p, c_i, c_p],
    "mc_argmax_step_f32": [c_p, c_l, c_p, c_p, c_l, c_p, c_i, c_i, c_p],
    "mc_llm_create": [C.POINTER(LlmConfigC), C.POINTER(c_p)],
    "mc_llm_destroy": [c_p],
    "mc_llm_set_weights": [c_p, C.POINTER(c_p), c_p, c_p, c_p, c_p, c_p],
    "mc_llm_set_option": [c_p, C.c_char_p, c_i],
    "mc_llm_set_sampling": [c_p, c_i, c_f, c_i, c_f],
    "mc_llm_set_key_mask": [c_p, c_p, c_l],
    "mc_sample_step_f32": [c_p, c_l, c_p, c_p, c_l, c_p, c_i, c_p, C.c_uint64, c_i, c_i, c_f, c_i, c_f, c_p, c_p, c_l, c_p],
    "mc_llm_workspace_bytes": [c_p, c_i, c_i, c_i, C.POINTER(c_l)],
    "mc_llm_prefill": [c_p, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p],
    "mc_llm_decode": [c_p, c_i, c_i, c_p, c_p, c_l, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p],
    "mc_llm_get_option": [c_p, C.c_char_p, C.POINTER(c_i)],
    "mc_llm_profile_kinds": [],
    "mc_ckpt_open": [C.c_char_p, C.POINTER(c_p)],
    "mc_ckpt_close": [c_p],
    "mc_ckpt_count": [c_p, C.POINTER(c_i)],
    "mc_ckpt_entry": [c_p, c_i, C.POINTER(C.c_char_p), C.POINTER(c_i), C.POINTER(c_i), C.POINTER(C.POINTER(c_l)), C.POINTER(C.POINTER(c_l)),
                      C.POINTER(c_p), C.POINTER(c_l)],
    "mc_ckpt_copy_to_device": [c_p, c_i, c_p, c_p],
    "mc_ckpt_entry_path": [c_p, c_i, C.POINTER(C.c_char_p)],
    "mc_ckpt_scalar_count": [c_p, C.POINTER(c_i)],
    "mc_ckpt_scalar": [c_p, c_i, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(c_i), C.POINTER(c_l), C.POINTER(C.c_double),
                       C.POINTER(c_p), C.POINTER(c_l)],
    "mc_llm_profile_read": [c_p, c_i, C.POINTER(C.c_double), C.POINTER(c_l)],
    "mc_llm_set_capture": [c_p, c_p, c_p],
    "mc_log_softmax_f32": [c_p, c_l, c_p, c_l, c_i, c_i, c_p],
    "mc_compose_multi_bf16": [C.POINTER(ComposeMultiArgsC), c_p],
    "mc_compose_batch_bf16": [C.POINTER(ComposeMultiArgsC), c_i, c_p],
    "mc_compose_retention_floats": [c_i, c_i, C.POINTER(c_l)],
    "mc_attn_probs_bf16": [c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_p, c_l, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p],
})
# diagnostic entry points of the PROBES build only (csrc/Makefile `probes` -> tools/probes/libmc_hip_probes.so, loaded with
# MC_PROBES_LIB=1): kernel A/B variants, forced tile shapes, clock stamps.  The shipped library does not export them.
_OPTIONAL: dict = {"mc_gemm_debug": [c_i], "mc_attn_debug": [c_i], "mc_gemm_clock_read": [c_i, C.POINTER(C.c_double)]}


class MCError(RuntimeError):
    pass


def exported_symbols():
    return sorted(list(_SIGS) + ["mc_last_error"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if _storage in _libs:
        _lib = _libs[_storage]
        return _lib
    LIB_PATH = LIB_PATHS[_storage]
    if not os.path.exists(LIB_PATH):
        raise MCError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"(or `make -C modelcompose_amd/csrc`). There is no CPU fallback.")
    # torch first: its wheel carries its own libamdhip64; if libmc_hip.so were loaded before it, the system HIP runtime would be bound
    # to our kernels and torch's copy to the tensors - two runtimes in one process, and our launches see "no ROCm-capable device"
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.mc_last_error.restype = C.c_char_p
    L.mc_last_error.argtypes = []
    for name, args in {**_SIGS, **_OPTIONAL}.items():
        fn = getattr(L, name, None)
        if fn is None:
            if name in _OPTIONAL:
                continue
            raise MCError(f"libmc_hip.so does not export {name}")
        fn.argtypes = args
        fn.restype = c_i
    v = L.mc_abi_version()
    if v != ABI_VERSION:
        raise MCError(f"{os.path.basename(LIB_PATH)} ABI version {v} != expected {ABI_VERSION}; rebuild")
    if L.mc_storage_dtype() != _DTYPE_CODES[_storage]:
        raise MCError(f"{os.path.basename(LIB_PATH)} was built for storage dtype code {L.mc_storage_dtype()}, expected {_storage}; rebuild")
    _libs[_storage] = L
    _lib = L
    return L


def check(status: int, what: str = ""):
    if status == 0:
        return
    msg = lib().mc_last_error().decode("utf-8", "replace")
    if status == 1:
        raise ValueError(f"{what}: {msg}")
    raise MCError(f"{what}: {msg}")
