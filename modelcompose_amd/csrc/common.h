// Shared device/host helpers for the gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

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
