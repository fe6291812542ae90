// Shared device/host helpers for the gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit STORAGE element of this build of the library.  libmc_hip.so: bfloat16 (BASELINE.json's dtype, the headline).  libmc_hip_f16.so:
// the same sources instantiated on IEEE half (-DMC_STORAGE_F16) - the reference's own inference dtype (modelcompose/model/builder.py:41,
// :162, :185: torch_dtype=torch.float16), 8x finer mantissa at the same MFMA rate (v_mfma_f32_16x16x32_f16): the parity instrument.  The
// name bf16_t / bf16xN is kept for the element type in both; every kernel converts through float, so nothing else depends on the format
// (the one bit-level routine, compose_round, has a form per format).
#ifdef MC_STORAGE_F16
typedef _Float16 bf16_t;
typedef _Float16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
#define MC_STORAGE_IS_F16 1
#define mc_mfma_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define mc_mfma_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#else
typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MC_STORAGE_IS_F16 0
#define mc_mfma_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define mc_mfma_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ds_read_b64_tr_b16 (the transposing LDS read: V / K / activation fragments straight from row-major tiles) for the build's storage element
typedef __attribute__((address_space(3))) void mc_lds_void;
#if MC_STORAGE_IS_F16
typedef __fp16 mc_tr4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ bf16x4 mc_ds_read_tr16(mc_lds_void* p) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) mc_tr4*)p));
}
#else
typedef __bf16 mc_tr4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
__device__ __forceinline__ bf16x4 mc_ds_read_tr16(mc_lds_void* p) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) mc_tr4*)p));
}
#endif

#define MC_WAVE 64

#include "../../include/mc_hip.h"   // activation codes, argument blocks

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

// 1 / (1 + e^-x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of the IEEE division sequence (~10 VALU instructions per element:
// measured +34 % on a CLIP fc1 launch with the QuickGELU epilogue, K = 1024).  Every caller rounds its result to bf16, 2^-15 coarser.
__device__ __forceinline__ float mc_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// exact-GELU 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 - every caller rounds to bf16, 2^-9 -
// and the tail x < 0 computed as the complement poly * e^-z^2 itself, so small results keep their relative accuracy): 1 v_rcp, 1 v_exp and
// 8 FMAs instead of ocml's branchy erff (a BEATs fc1 launch: +47 % over the same launch without activation)
__device__ __forceinline__ float mc_gelu(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float pe = poly * __expf(-z * z);                 // = erfc(z)
    return 0.5f * x * (x >= 0.f ? 2.0f - pe : pe);
}

__device__ __forceinline__ float mc_act(float x, int act) {
    switch (act) {
        case MC_ACT_GELU: return mc_gelu(x);
        case MC_ACT_QUICK_GELU: return x * mc_sigmoid(1.702f * x);
        case MC_ACT_SILU: return x * mc_sigmoid(x);
        case MC_ACT_RELU: return fmaxf(x, 0.0f);
        default: return x;
    }
}

// RoPE of one pair (a, b) = (x[d], x[d + D/2]) with cos / sin of (position, d) (apply_rotary_pos_emb, multimodal_llama.py:281-295: x cos +
// rotate_half(x) sin).  Explicit FMAs: rope_kv_kernel and the q|k|v GEMM epilogue that replaces it round identically.
__device__ __forceinline__ void mc_rope_pair(float a, float b, float c, float s, float& o1, float& o2) {
    o1 = fmaf(a, c, -(b * s));
    o2 = fmaf(b, c, a * s);
}

// Philox4x32-10 (Salmon et al. 2011; constants of the Random123 distribution, pinned by its known-answer vectors in tests/test_oracle_golden.py)
__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Dropout on attention probabilities (BertSelfAttention, multimodal_projector/Qformer.py:259: self.dropout(attention_probs)): element
// e = ((b H + h) Lq + q) S + key is kept iff word (e & 3) of Philox(counter = (e >> 2 lo, e >> 2 hi, stream_id, 0), key = seed) >= thr - the
// mask rule of mc_dropout_bf16 over the [B H Lq, S] score matrix, regenerated by the forward and both backward kernels (S % 4 == 0).
struct AttnDropout { uint32_t thr, seed_lo, seed_hi, stream_id; float inv_keep; };
// the 4 keep factors (0 or 1 / (1 - p)) of keys key4 .. key4 + 3 (key4 % 4 == 0) of query row `qrow` = (b H + h) Lq + q
__device__ __forceinline__ void attn_dropout_quad(const AttnDropout& d, int64_t qrow, int S, int key4, float (&m)[4]) {
    const uint64_t e4 = ((uint64_t)qrow * (uint64_t)S + (uint64_t)key4) >> 2;
    uint32_t r[4];
    philox4((uint32_t)e4, (uint32_t)(e4 >> 32), d.stream_id, 0u, d.seed_lo, d.seed_hi, r);
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = r[j] >= d.thr ? d.inv_keep : 0.f;
}
// one element
__device__ __forceinline__ float attn_dropout_one(const AttnDropout& d, int64_t qrow, int S, int key) {
    float m[4];
    attn_dropout_quad(d, qrow, S, key & ~3, m);
    return m[key & 3];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024; `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// host-side error plumbing (capi.cpp owns the storage)
void mc_set_error(const char* fmt, ...);
#define MC_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            mc_set_error(__VA_ARGS__);   \
            return 1;                    \
        }                                \
    } while (0)
#define MC_CHECK_LAUNCH()                                                   \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            mc_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return 2;                                                       \
        }                                                                   \
    } while (0)
