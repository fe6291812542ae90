// Weight composition ("online-merge-reset" made dense):  W' = W + sum_m s_m * B_m * A_m
//
// The reference composes at every forward inside LocalLoraLinear.forward
// (modelcompose/model/language_model/multimodal_llama.py:130-157: y = xW^T + sum_m s_m B_m(A_m x), with
//  s_m = lora_alpha/r * coefficient from reset_scaling_weights, :92-107).  The dense form is the reference's own
// get_delta_weight (scripts/evaluate_delta_weights.py:8-15:  dW = (B @ A) * alpha/r).  One pass reads W once
// (row-major bf16 as stored in the checkpoint), accumulates every modality's rank-r product on the MFMA
// units in fp32, applies the fp32 coefficients, rounds ONCE to bf16 and writes the MFMA-packed layout used
// by gemm.hip — a fused multi-term AXPY over the state_dict tensors.
//
// Operands: Bm [N, r] row-major (checkpoint layout of lora_B), At [K, r] row-major (= lora_A transposed once
// on load; r contiguous so both fragments are 16-byte loads).
#include "common.h"

#define MC_MAX_TERMS 8

struct ComposeParams {
    const bf16_t* w; int64_t ldw;            // may be null (-> pure delta)
    const bf16_t* at[MC_MAX_TERMS];          // [K, r]
    const bf16_t* bm[MC_MAX_TERMS];          // [N, r]
    float scale[MC_MAX_TERMS];
    int n_terms, r;
    bf16_t* out_packed;                      // [ceil16(N)/16][Kp/32][64][8]
    bf16_t* out_rowmajor; int64_t ldo;       // optional row-major copy (debug / parity), may be null
    int N, K, Kp;
    const float* col_scale;                  // [K] fp32 or null: W'[n][k] *= col_scale[k] before the single bf16 rounding
    int nb_stride, nb_offset;                // packed 16-row block nb is written at block index nb*nb_stride + nb_offset
    float* retention_parts;                  // optional [gridDim.y][gridDim.x][2]: per-workgroup partial sums of (W' - bf16(W c)) * (dW c) and
                                             // (dW c)^2: how much of the delta survives the single bf16 rounding (see mc_hip.h)
    uint32_t dither_seed;                    // 0: round to nearest even.  != 0: UNBIASED rounding (round up with probability = the discarded
                                             // fraction, from a counter hash of (seed, n, k)): E[W'] = the fp32 value, so a delta below half a
                                             // bf16 step of W is kept in expectation instead of rounding back to W (round 4; mc_hip.h)
};

// 32-bit mix (lowbias32): the dither's uniform bits for element (n, k)
__device__ __forceinline__ uint32_t compose_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ bf16_t compose_round(float v, uint32_t seed, int n, int k) {
    if (seed == 0) return (bf16_t)v;
    uint32_t bits = __builtin_bit_cast(uint32_t, v);
    if ((bits & 0x7f800000U) == 0x7f800000U || (bits & 0x7fff0000U) == 0x7f7f0000U) return (bf16_t)v;   // inf / nan, and the largest finite bf16 (the carry would make it inf): as the plain cast
    const uint32_t r = compose_hash(seed ^ compose_hash((uint32_t)n * 0x9E3779B1U + (uint32_t)k)) & 0xFFFFU;
    bits += r;                                                                // sign-magnitude: the magnitude goes up with probability frac / 2^16
    const uint16_t hi = (uint16_t)(bits >> 16);
    return __builtin_bit_cast(bf16_t, hi);
}

// workgroup = 4 waves; tile = 32 rows (n) x 256 cols (k); wave w owns cols [64w, 64w+64)
__global__ __launch_bounds__(256) void compose_kernel(ComposeParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * 32;
    const int k0 = blockIdx.x * 256 + wave * 64;
    const bool wave_on = k0 < p.Kp;
    float ret_num = 0.f, ret_den = 0.f;
    if (wave_on) {

    f32x4 tot[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) tot[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int m = 0; m < p.n_terms; ++m) {
        f32x4 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bf16_t* at = p.at[m];
        const bf16_t* bm = p.bm[m];
        for (int rs = 0; rs < p.r; rs += 32) {
            bf16x8 bf[2], af[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = min(n0 + i * 16 + c16, p.N - 1);
                bf[i] = *(const bf16x8*)(bm + (int64_t)n * p.r + rs + g * 8);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = min(k0 + t * 16 + c16, p.K - 1);
                af[t] = *(const bf16x8*)(at + (int64_t)k * p.r + rs + g * 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bf[i], acc[i][t], 0, 0, 0);
        }
        const float s = p.scale[m];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) tot[i][t] += acc[i][t] * s;
    }
    // D[k = 4g + reg][n = c16]
    const int kblocks = p.Kp >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + i * 16 + c16;
        const int nb = n >> 4;
        if (nb * 16 >= ((p.N + 15) & ~15)) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = k0 + t * 16 + g * 4;
            float r4[4] = {tot[i][t][0], tot[i][t][1], tot[i][t][2], tot[i][t][3]};
            const bool inb = n < p.N;
            if (p.w && inb) {
                if (k + 3 < p.K) {
                    const bf16x4 w4 = *(const bf16x4*)(p.w + (int64_t)n * p.ldw + k);
#pragma unroll
                    for (int j = 0; j < 4; ++j) r4[j] += (float)w4[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < p.K) r4[j] += (float)p.w[(int64_t)n * p.ldw + k + j];
                }
            }
            if (p.col_scale) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k + j < p.K) r4[j] *= p.col_scale[k + j];
            }
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (inb && k + j < p.K) ? compose_round(r4[j], p.dither_seed, n, k + j) : (bf16_t)0.0f;
            if (p.retention_parts && p.w && inb) {
                // the composed weight against the base weight rounded the same way: the part of (W' - bf16(W c)) that lies along dW c
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k + j < p.K) {
                        const float cs = p.col_scale ? p.col_scale[k + j] : 1.0f;
                        const float wv = (float)p.w[(int64_t)n * p.ldw + k + j];
                        const float d = tot[i][t][j] * cs;
                        const float moved = (float)o[j] - (float)(bf16_t)(wv * cs);
                        ret_num = fmaf(moved, d, ret_num);
                        ret_den = fmaf(d, d, ret_den);
                    }
            }
            const int kb = k >> 5;
            const int q = (k & 31) >> 3;
            bf16_t* dst = p.out_packed + ((int64_t)(nb * p.nb_stride + p.nb_offset) * kblocks + kb) * 512 + ((q << 4) | (n & 15)) * 8 + (k & 7);
            *(bf16x4*)dst = o;
            if (p.out_rowmajor && inb) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k + j < p.K) p.out_rowmajor[(int64_t)n * p.ldo + k + j] = o[j];
            }
        }
    }
    }
    if (p.retention_parts) {
        __shared__ float red[4][2];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ret_num += __shfl_xor(ret_num, o, 64);
            ret_den += __shfl_xor(ret_den, o, 64);
        }
        if (lane == 0) { red[wave][0] = ret_num; red[wave][1] = ret_den; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float* dst = p.retention_parts + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
            dst[0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);            // fixed order: reproducible
            dst[1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        }
    }
}

extern "C" int mc_compose_weight_dither_bf16(const void* w, int64_t ldw, const void* const* at_list, const void* const* b_list,
                                             const float* scales, int n_terms, int r, void* out_packed, void* out_rowmajor,
                                             int64_t ldo, int N, int K, const float* col_scale, int nb_stride, int nb_offset,
                                             float* retention_parts, uint32_t dither_seed, void* stream) {
    MC_CHECK_ARG(out_packed && N > 0 && K > 0, "mc_compose_weight_bf16: bad arguments");
    MC_CHECK_ARG(n_terms >= 0 && n_terms <= MC_MAX_TERMS, "mc_compose_weight_bf16: at most %d terms (got %d)", MC_MAX_TERMS, n_terms);
    MC_CHECK_ARG(n_terms == 0 || (r > 0 && r % 32 == 0), "mc_compose_weight_bf16: rank %d must be a multiple of 32 (pad A^T / B)", r);
    MC_CHECK_ARG(!w || ldw % 4 == 0, "mc_compose_weight_bf16: ldw must be a multiple of 4");
    MC_CHECK_ARG(nb_stride >= 1 && nb_offset >= 0 && nb_offset < nb_stride, "mc_compose_weight_ex_bf16: bad block interleave %d/%d", nb_offset, nb_stride);
    ComposeParams p;
    p.w = (const bf16_t*)w; p.ldw = ldw;
    for (int i = 0; i < n_terms; ++i) {
        MC_CHECK_ARG(at_list[i] && b_list[i], "mc_compose_weight_bf16: null term %d", i);
        p.at[i] = (const bf16_t*)at_list[i]; p.bm[i] = (const bf16_t*)b_list[i]; p.scale[i] = scales[i];
    }
    p.n_terms = n_terms; p.r = r;
    p.out_packed = (bf16_t*)out_packed; p.out_rowmajor = (bf16_t*)out_rowmajor; p.ldo = ldo;
    p.N = N; p.K = K; p.Kp = (K + 63) / 64 * 64;
    p.col_scale = col_scale; p.nb_stride = nb_stride; p.nb_offset = nb_offset;
    p.retention_parts = retention_parts;
    p.dither_seed = dither_seed;
    dim3 grid((p.Kp + 255) / 256, (N + 31) / 32);
    compose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
    MC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mc_compose_weight_ex_bf16(const void* w, int64_t ldw, const void* const* at_list, const void* const* b_list,
                                         const float* scales, int n_terms, int r, void* out_packed, void* out_rowmajor,
                                         int64_t ldo, int N, int K, const float* col_scale, int nb_stride, int nb_offset,
                                         float* retention_parts, void* stream) {
    return mc_compose_weight_dither_bf16(w, ldw, at_list, b_list, scales, n_terms, r, out_packed, out_rowmajor, ldo, N, K, col_scale, nb_stride,
                                         nb_offset, retention_parts, 0u, stream);
}

extern "C" int mc_compose_weight_bf16(const void* w, int64_t ldw, const void* const* at_list, const void* const* b_list,
                                      const float* scales, int n_terms, int r, void* out_packed, void* out_rowmajor,
                                      int64_t ldo, int N, int K, void* stream) {
    return mc_compose_weight_ex_bf16(w, ldw, at_list, b_list, scales, n_terms, r, out_packed, out_rowmajor, ldo, N, K, nullptr, 1, 0,
                                     nullptr, stream);
}
