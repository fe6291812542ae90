// HBM-bound row kernels (bf16 storage, fp32 math, 16-byte vector accesses).
//
//   rmsnorm      – LlamaRMSNorm (transformers==4.31; multimodal_llama.py:405-406,433,455,603)
//   layernorm    – nn.LayerNorm of the CLIP / encoder blocks
//   rope_kv      – apply_rotary_pos_emb (rotate-half pairs (i, i+D/2)) fused with the KV-cache append that
//                  replaces torch.cat (multimodal_llama.py:281-289) and with the routed->sequence re-ordering
//   silu_mul     – act_fn(gate) * up (multimodal_llama.py:392-394)
//   copy_rows    – row gather/scatter used by the splice (multimodal_arch.py:349-378) and embed_tokens
//   argmax       – greedy next-token selection (transformers greedy_search; model_multimodal_qa_loader.py:94-102)
//   im2col       – patch-embed conv as a GEMM operand (CLIP conv14/s14 etc.)
//   vit_assemble – class token + learned positions
#include "common.h"

// ------------------------------------------------------------------------------------------
template <bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ w,
                                                   const bf16_t* __restrict__ bias, bf16_t* __restrict__ out, int64_t ldo,
                                                   int D, float eps) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const bf16_t* xr = x + (int64_t)row * ldx;
    bf16_t* orow = out + (int64_t)row * ldo;
    const int nv = D >> 3;
    // up to 4 vectors of 8 per thread cached in registers (D <= 8192); larger rows re-read
    float v[4][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = it * 256 + threadIdx.x;
        if (i < nv) {
            const bf16x8 t = *(const bf16x8*)(xr + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = (float)t[j];
                v[it][j] = f;
                s1 += f;
                s2 += f * f;
            }
        }
    }
    for (int i = 1024 + threadIdx.x; i < nv; i += 256) {
        const bf16x8 t = *(const bf16x8*)(xr + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)t[j]; s1 += f; s2 += f * f; }
    }
    float mean = 0.f, rstd;
    if (RMS) {
        s2 = block_sum(s2, red);
        rstd = rsqrtf(s2 / D + eps);
    } else {
        s1 = block_sum(s1, red);
        mean = s1 / D;
        // two-pass variance on the cached values for accuracy
        float sv = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = it * 256 + threadIdx.x;
            if (i < nv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[it][j] - mean; sv += d * d; }
            }
        }
        for (int i = 1024 + threadIdx.x; i < nv; i += 256) {
            const bf16x8 t = *(const bf16x8*)(xr + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = (float)t[j] - mean; sv += d * d; }
        }
        sv = block_sum(sv, red);
        rstd = rsqrtf(sv / D + eps);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = it * 256 + threadIdx.x;
        if (i < nv) {
            const bf16x8 wv = *(const bf16x8*)(w + i * 8);
            bf16x8 o;
            if (bias) {
                const bf16x8 bv = *(const bf16x8*)(bias + i * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((v[it][j] - mean) * rstd * (float)wv[j] + (float)bv[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((v[it][j] - mean) * rstd * (float)wv[j]);
            }
            *(bf16x8*)(orow + i * 8) = o;
        }
    }
    for (int i = 1024 + threadIdx.x; i < nv; i += 256) {
        const bf16x8 t = *(const bf16x8*)(xr + i * 8);
        const bf16x8 wv = *(const bf16x8*)(w + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float r = ((float)t[j] - mean) * rstd * (float)wv[j];
            if (bias) r += (float)bias[i * 8 + j];
            o[j] = (bf16_t)r;
        }
        *(bf16x8*)(orow + i * 8) = o;
    }
}

// Rows of at most 2048 elements (the encoders' LayerNorms: 768 / 1024 / 1408 wide): one WAVE per row, four rows per workgroup - a row is
// 2-4 loads of 16 bytes per lane, the statistics are wave reductions (no LDS, no barrier), and the launch has a quarter of the
// workgroups.  Same two-pass variance on the cached values as norm_kernel.  (norm_kernel spent a 256-thread workgroup with two block
// reductions on a 2-KiB row: 114 us for the video tower's 98.7 K x 1024 activations = 3.5 TB/s.)
template <bool RMS>
__global__ __launch_bounds__(256) void norm_rows_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ bias, bf16_t* __restrict__ out, int64_t ldo, int M,
                                                        int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* xr = x + (int64_t)row * ldx;
    bf16_t* orow = out + (int64_t)row * ldo;
    const int nv = D >> 3;
    float v[4][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = it * 64 + lane;
        if (i < nv) {
            const bf16x8 t = *(const bf16x8*)(xr + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = (float)t[j];
                v[it][j] = f;
                s1 += f;
                s2 += f * f;
            }
        }
    }
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(wave_sum(s2) / D + eps);
    } else {
        mean = wave_sum(s1) / D;
        float sv = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (it * 64 + lane < nv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[it][j] - mean; sv += d * d; }
            }
        }
        rstd = rsqrtf(wave_sum(sv) / D + eps);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = it * 64 + lane;
        if (i < nv) {
            const bf16x8 wv = *(const bf16x8*)(w + i * 8);
            bf16x8 o;
            if (bias) {
                const bf16x8 bv = *(const bf16x8*)(bias + i * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((v[it][j] - mean) * rstd * (float)wv[j] + (float)bv[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((v[it][j] - mean) * rstd * (float)wv[j]);
            }
            *(bf16x8*)(orow + i * 8) = o;
        }
    }
}

extern "C" int mc_rmsnorm_bf16(const void* x, int64_t ldx, const void* w, void* out, int64_t ldo, int M, int D, float eps,
                               void* stream) {
    MC_CHECK_ARG(x && w && out && M > 0 && D > 0 && D % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "mc_rmsnorm_bf16: bad arguments (D=%d)", D);
    if (D <= 2048) norm_rows_kernel<true><<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (const bf16_t*)w, nullptr, (bf16_t*)out, ldo, M, D, eps);
    else norm_kernel<true><<<M, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (const bf16_t*)w, nullptr, (bf16_t*)out, ldo, D, eps);
    MC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mc_layernorm_bf16(const void* x, int64_t ldx, const void* w, const void* b, void* out, int64_t ldo, int M,
                                 int D, float eps, void* stream) {
    MC_CHECK_ARG(x && w && out && M > 0 && D > 0 && D % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "mc_layernorm_bf16: bad arguments (D=%d)", D);
    if (D <= 2048) norm_rows_kernel<false><<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out, ldo, M, D, eps);
    else norm_kernel<false><<<M, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out, ldo, D, eps);
    MC_CHECK_LAUNCH();
    return 0;
}

// h[r] += table[idx[r]] and LayerNorm of the new row in ONE pass (round 4; LanguageBind-Video's temporal branch, video/modeling_video.py:105-115:
// `hidden_states + temporal_embedding[:, :t]` then temporal_layer_norm1).  One wave per row like norm_rows_kernel; the sum is rounded to
// bf16 and stored (it is the residual the branch's output is added to), and the statistics are taken over those bf16 values - exactly
// what add_rows_kernel followed by norm_rows_kernel computes, in one read and two writes instead of two reads and ... three passes.
__global__ __launch_bounds__(256) void add_norm_rows_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ table, int64_t ldt,
                                                            const int32_t* __restrict__ idx, bf16_t* __restrict__ sum_out, int64_t lds,
                                                            const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias, bf16_t* __restrict__ out,
                                                            int64_t ldo, int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* xr = x + (int64_t)row * ldx;
    const bf16_t* tr = table + (int64_t)(idx ? idx[row] : row) * ldt;
    bf16_t* sr = sum_out + (int64_t)row * lds;
    bf16_t* orow = out + (int64_t)row * ldo;
    const int nv = D >> 3;
    float v[4][8];
    float s1 = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = it * 64 + lane;
        if (i < nv) {
            const bf16x8 a = *(const bf16x8*)(xr + i * 8), b = *(const bf16x8*)(tr + i * 8);
            bf16x8 sm;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sm[j] = (bf16_t)((float)a[j] + (float)b[j]);
                const float f = (float)sm[j];
                v[it][j] = f;
                s1 += f;
            }
            *(bf16x8*)(sr + i * 8) = sm;
        }
    }
    const float mean = wave_sum(s1) / D;
    float sv = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (it * 64 + lane < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[it][j] - mean; sv += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sv) / D + eps);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = it * 64 + lane;
        if (i < nv) {
            const bf16x8 wv = *(const bf16x8*)(w + i * 8), bv = *(const bf16x8*)(bias + i * 8);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((v[it][j] - mean) * rstd * (float)wv[j] + (float)bv[j]);
            *(bf16x8*)(orow + i * 8) = o;
        }
    }
}

extern "C" int mc_add_layernorm_bf16(const void* x, int64_t ldx, const void* table, int64_t ldt, const int32_t* idx, void* sum_out, int64_t lds,
                                     const void* w, const void* b, void* out, int64_t ldo, int M, int D, float eps, void* stream) {
    MC_CHECK_ARG(x && table && sum_out && w && b && out && M > 0 && D > 0 && D % 8 == 0 && D <= 2048 && ldx % 8 == 0 && ldt % 8 == 0 && lds % 8 == 0 &&
                 ldo % 8 == 0, "mc_add_layernorm_bf16: bad arguments (D=%d)", D);
    add_norm_rows_kernel<<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (const bf16_t*)table, ldt, idx, (bf16_t*)sum_out, lds,
                                                                       (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)out, ldo, M, D, eps);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// RMSNorm split in two: the norm weight is folded into the next linear's columns at compose time, so what is left at run
// time is the per-row factor 1/rms (LlamaRMSNorm: variance in fp32 of the bf16 hidden state, multimodal_llama.py:405-406),
// applied by the GEMM epilogue (Epilogue::row_scale).
//   rms_scale_kernel     : rs[m] = rsqrt(mean(x[m]^2) + eps)                                  (prefill, after embedding)
__global__ __launch_bounds__(256) void rms_scale_kernel(const bf16_t* __restrict__ x, int64_t ldx, float* __restrict__ rs, int D, float eps) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const bf16_t* xr = x + (int64_t)row * ldx;
    float s2 = 0.f;
    for (int i = threadIdx.x; i < (D >> 3); i += 256) {
        const bf16x8 t = *(const bf16x8*)(xr + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = (float)t[j]; s2 += f * f; }
    }
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) rs[row] = rsqrtf(s2 / D + eps);
}

extern "C" int mc_rms_scale_bf16(const void* x, int64_t ldx, float* row_scale, int M, int D, float eps, void* stream) {
    MC_CHECK_ARG(x && row_scale && M > 0 && D > 0 && D % 8 == 0 && ldx % 8 == 0, "mc_rms_scale_bf16: bad arguments (D=%d)", D);
    rms_scale_kernel<<<M, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, row_scale, D, eps);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// RoPE + routed->sequence scatter + KV-cache append.
// qkv row r (routed order) = [q (H*D) | k (Hkv*D) | v (Hkv*D)]; row_b[r] batch entry, row_pos[r] absolute position,
// row_t[r] query index inside this call (0..Lq-1).
// q_out[(b*Lq + t)*H*D + h*D + d];  k/v cache [B][Hkv][Smax][D].
__global__ __launch_bounds__(256) void rope_kv_kernel(const bf16_t* __restrict__ qkv, int64_t ld, const int32_t* __restrict__ row_b,
                                                      const int32_t* __restrict__ row_pos, const int32_t* __restrict__ row_t,
                                                      const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                      bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_cache,
                                                      bf16_t* __restrict__ v_cache, int H, int Hkv, int D, int Lq, int Smax) {
    const int r = blockIdx.x;
    const int b = row_b[r], pos = row_pos[r], t = row_t[r];
    if (b < 0) return;
    const int half = D >> 1, cpd = half >> 3;            // 8-wide chunks per half head
    const bf16_t* src = qkv + (int64_t)r * ld;
    const float* cr = cosT + (int64_t)pos * half;
    const float* sr = sinT + (int64_t)pos * half;
    const int n_rot = (H + Hkv) * cpd;
    const int n_all = n_rot + Hkv * cpd;
    for (int it = threadIdx.x; it < n_all; it += 256) {
        if (it < n_rot) {
            const int hh = it / cpd, ch = it % cpd;
            const bf16_t* s0 = src + hh * D + ch * 8;
            const bf16x8 x1 = *(const bf16x8*)s0, x2 = *(const bf16x8*)(s0 + half);
            bf16x8 o1, o2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float c = cr[ch * 8 + j], s = sr[ch * 8 + j];
                const float a = (float)x1[j], bb = (float)x2[j];
                float r1, r2;
                mc_rope_pair(a, bb, c, s, r1, r2);
                o1[j] = (bf16_t)r1;
                o2[j] = (bf16_t)r2;
            }
            bf16_t* dst;
            if (hh < H) dst = q_out + ((int64_t)(b * Lq + t) * H + hh) * D + ch * 8;
            else dst = k_cache + (((int64_t)b * Hkv + (hh - H)) * Smax + pos) * D + ch * 8;
            *(bf16x8*)dst = o1;
            *(bf16x8*)(dst + half) = o2;
        } else {
            const int i2 = it - n_rot;
            const int hh = i2 / cpd, ch = i2 % cpd;
            const bf16_t* s0 = src + (H + Hkv + hh) * D + ch * 8;
            bf16_t* dst = v_cache + (((int64_t)b * Hkv + hh) * Smax + pos) * D + ch * 8;
            *(bf16x8*)dst = *(const bf16x8*)s0;
            *(bf16x8*)(dst + half) = *(const bf16x8*)(s0 + half);
        }
    }
}

extern "C" int mc_rope_kv_bf16(const void* qkv, int64_t ld, const int32_t* row_b, const int32_t* row_pos,
                               const int32_t* row_t, const float* cos_table, const float* sin_table, void* q_out,
                               void* k_cache, void* v_cache, int M, int H, int Hkv, int D, int Lq, int Smax, void* stream) {
    MC_CHECK_ARG(qkv && row_b && row_pos && row_t && cos_table && sin_table && q_out && k_cache && v_cache, "mc_rope_kv_bf16: null pointer");
    MC_CHECK_ARG(M > 0 && D % 16 == 0 && ld % 8 == 0, "mc_rope_kv_bf16: bad shape D=%d ld=%lld", D, (long long)ld);
    rope_kv_kernel<<<M, 256, 0, (hipStream_t)stream>>>((const bf16_t*)qkv, ld, row_b, row_pos, row_t, cos_table, sin_table,
                                                       (bf16_t*)q_out, (bf16_t*)k_cache, (bf16_t*)v_cache, H, Hkv, D, Lq, Smax);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silu_mul_kernel(const bf16_t* __restrict__ gu, int64_t ld, bf16_t* __restrict__ out,
                                                       int64_t ldo, int M, int I) {
    const int nv = I >> 3;
    const int64_t total = (int64_t)M * nv;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / nv), c = (int)(i % nv);
        const bf16x8 g = *(const bf16x8*)(gu + (int64_t)m * ld + c * 8);
        const bf16x8 u = *(const bf16x8*)(gu + (int64_t)m * ld + I + c * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gf = (float)g[j];
            o[j] = (bf16_t)(gf * mc_sigmoid(gf) * (float)u[j]);
        }
        *(bf16x8*)(out + (int64_t)m * ldo + c * 8) = o;
    }
}

extern "C" int mc_silu_mul_bf16(const void* gate_up, int64_t ld, void* out, int64_t ldo, int M, int I, void* stream) {
    MC_CHECK_ARG(gate_up && out && M > 0 && I > 0 && I % 8 == 0 && ld % 8 == 0 && ldo % 8 == 0, "mc_silu_mul_bf16: bad arguments");
    const int64_t total = (int64_t)M * (I >> 3);
    const int grid = (int)min((int64_t)8192, (total + 255) / 256);
    silu_mul_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)gate_up, ld, (bf16_t*)out, ldo, M, I);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// dst[dst_idx[i]] = src[src_idx[i]]  (rows of D bf16; idx null -> i; negative -> skip / zero row)
template <typename IdxT>
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_t* __restrict__ src, int64_t lds_, const IdxT* __restrict__ src_idx,
                                                        bf16_t* __restrict__ dst, int64_t ldd, const int32_t* __restrict__ dst_idx,
                                                        int D) {
    const int i = blockIdx.x;
    const int64_t si = src_idx ? (int64_t)src_idx[i] : i;
    const int64_t di = dst_idx ? (int64_t)dst_idx[i] : i;
    if (di < 0) return;
    const int nv = D >> 3;
    bf16_t* d = dst + di * ldd;
    if (si < 0) {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = threadIdx.x; c < nv; c += 256) *(bf16x8*)(d + c * 8) = z;
        return;
    }
    const bf16_t* s = src + si * lds_;
    for (int c = threadIdx.x; c < nv; c += 256) *(bf16x8*)(d + c * 8) = *(const bf16x8*)(s + c * 8);
}

// dst[b] = src[b * Lq + lens[b] - 1]: the last valid row of every sequence of a [B, Lq, D] tensor (rows of D bf16)
__global__ __launch_bounds__(256) void gather_last_rows_kernel(const bf16_t* __restrict__ src, int64_t lds_, const int32_t* __restrict__ lens, int Lq,
                                                               bf16_t* __restrict__ dst, int64_t ldd, int D) {
    const int b = blockIdx.x;
    const int t = min(max(lens[b], 1), Lq) - 1;
    const bf16_t* s = src + ((int64_t)b * Lq + t) * lds_;
    bf16_t* d = dst + (int64_t)b * ldd;
    for (int c = threadIdx.x; c < (D >> 3); c += 256) *(bf16x8*)(d + c * 8) = *(const bf16x8*)(s + c * 8);
}

extern "C" int mc_gather_last_rows_bf16(const void* src, int64_t ld_src, const int32_t* lens, int Lq, void* dst, int64_t ld_dst, int B, int D,
                                        void* stream) {
    MC_CHECK_ARG(src && lens && dst && B > 0 && Lq > 0 && D > 0 && D % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "mc_gather_last_rows_bf16: bad arguments");
    gather_last_rows_kernel<<<B, 256, 0, (hipStream_t)stream>>>((const bf16_t*)src, ld_src, lens, Lq, (bf16_t*)dst, ld_dst, D);
    MC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mc_copy_rows_bf16(const void* src, int64_t ld_src, const int32_t* src_idx, void* dst, int64_t ld_dst,
                                 const int32_t* dst_idx, int n_rows, int D, void* stream) {
    MC_CHECK_ARG(src && dst && n_rows >= 0 && D > 0 && D % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "mc_copy_rows_bf16: bad arguments");
    if (n_rows == 0) return 0;
    copy_rows_kernel<int32_t><<<n_rows, 256, 0, (hipStream_t)stream>>>((const bf16_t*)src, ld_src, src_idx, (bf16_t*)dst, ld_dst, dst_idx, D);
    MC_CHECK_LAUNCH();
    return 0;
}

// embedding gather with int64 token ids (embed_tokens)
extern "C" int mc_embed_rows_bf16(const void* table, int64_t ld_table, const int64_t* ids, void* dst, int64_t ld_dst,
                                  const int32_t* dst_idx, int n_rows, int D, void* stream) {
    MC_CHECK_ARG(table && ids && dst && n_rows >= 0 && D % 8 == 0, "mc_embed_rows_bf16: bad arguments");
    if (n_rows == 0) return 0;
    copy_rows_kernel<int64_t><<<n_rows, 256, 0, (hipStream_t)stream>>>((const bf16_t*)table, ld_table, ids, (bf16_t*)dst, ld_dst, dst_idx, D);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// argmax over fp32 rows; ties -> lowest index (torch.argmax CPU behaviour). Writes int64 ids.
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ x, int64_t ld, int64_t* __restrict__ out, int N) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const float* r = x + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float v = r[i];
        if (v > best || (v == best && i < idx)) { best = v; idx = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64);
        const int i2 = __shfl_xor(idx, o, 64);
        if (v2 > best || (v2 == best && i2 < idx)) { best = v2; idx = i2; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[blockIdx.x] = idx;
    }
}

extern "C" int mc_argmax_f32(const void* x, int64_t ld, int64_t* out, int M, int N, void* stream) {
    MC_CHECK_ARG(x && out && M > 0 && N > 0, "mc_argmax_f32: bad arguments");
    argmax_kernel<<<M, 256, 0, (hipStream_t)stream>>>((const float*)x, ld, out, N);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// im2col for a strided 2-D conv without padding: in [B, C, Hin, Win] (bf16) -> out [B*oh*ow, Kp],
// column = (c*kh + i)*kw + j (conv weight [Cout, C, kh, kw] flattened), zero padded to Kp.
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int C, int Hin,
                                                     int Win, int kh, int kw, int sh, int sw, int oh, int ow, int Kp) {
    const int K = C * kh * kw;
    const int64_t total = (int64_t)B * oh * ow * Kp;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % Kp);
        const int64_t row = i / Kp;
        bf16_t v = (bf16_t)0.0f;
        if (col < K) {
            const int j = col % kw, ii = (col / kw) % kh, c = col / (kw * kh);
            const int ox = (int)(row % ow), oy = (int)((row / ow) % oh), b = (int)(row / ((int64_t)ow * oh));
            v = in[(((int64_t)b * C + c) * Hin + oy * sh + ii) * Win + ox * sw + j];
        }
        out[i] = v;
    }
}

extern "C" int mc_im2col_bf16(const void* in, void* out, int B, int C, int Hin, int Win, int kh, int kw, int sh, int sw,
                              int Kp, void* stream) {
    MC_CHECK_ARG(in && out && B > 0 && C > 0 && Hin >= kh && Win >= kw && sh > 0 && sw > 0 && Kp >= C * kh * kw, "mc_im2col_bf16: bad arguments");
    const int oh = (Hin - kh) / sh + 1, ow = (Win - kw) / sw + 1;
    const int64_t total = (int64_t)B * oh * ow * Kp;
    const int grid = (int)min((int64_t)16384, (total + 255) / 256);
    im2col_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)in, (bf16_t*)out, B, C, Hin, Win, kh, kw, sh, sw, oh, ow, Kp);
    MC_CHECK_LAUNCH();
    return 0;
}

// out[b, 0] = cls + pos[0]; out[b, 1+t] = patches[b*T + t] + pos[1+t]   (cls may be null: out[b,t] = patches + pos[t])
__global__ __launch_bounds__(256) void vit_assemble_kernel(const bf16_t* __restrict__ patches, const bf16_t* __restrict__ cls,
                                                           const bf16_t* __restrict__ pos, bf16_t* __restrict__ out, int T, int D) {
    const int tok = blockIdx.x, b = blockIdx.y;
    const int has_cls = cls != nullptr;
    const int Tt = T + has_cls;
    const bf16_t* src = (has_cls && tok == 0) ? cls : patches + ((int64_t)b * T + tok - has_cls) * D;
    const bf16_t* pr = pos ? pos + (int64_t)tok * D : nullptr;
    bf16_t* dst = out + ((int64_t)b * Tt + tok) * D;
    for (int c = threadIdx.x; c < (D >> 3); c += 256) {
        const bf16x8 a = *(const bf16x8*)(src + c * 8);
        bf16x8 o;
        if (pr) {
            const bf16x8 pp = *(const bf16x8*)(pr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)a[j] + (float)pp[j]);
        } else o = a;
        *(bf16x8*)(dst + c * 8) = o;
    }
}

extern "C" int mc_vit_assemble_bf16(const void* patches, const void* cls, const void* pos, void* out, int B, int T, int D,
                                    void* stream) {
    MC_CHECK_ARG(patches && out && B > 0 && T > 0 && D % 8 == 0, "mc_vit_assemble_bf16: bad arguments");
    dim3 grid(T + (cls ? 1 : 0), B);
    vit_assemble_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)patches, (const bf16_t*)cls, (const bf16_t*)pos, (bf16_t*)out, T, D);
    MC_CHECK_LAUNCH();
    return 0;
}

// generic elementwise: out = a + b (bf16, same shape, contiguous)
__global__ __launch_bounds__(256) void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, int64_t nv) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
        const bf16x8 x = *(const bf16x8*)(a + i * 8), y = *(const bf16x8*)(b + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)x[j] + (float)y[j]);
        *(bf16x8*)(out + i * 8) = o;
    }
}

extern "C" int mc_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
    MC_CHECK_ARG(a && b && out && n > 0 && n % 8 == 0, "mc_add_bf16: bad arguments");
    const int64_t nv = n >> 3;
    const int grid = (int)min((int64_t)8192, (nv + 255) / 256);
    add_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, nv);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// device-resident greedy-loop state (so that every decode step is the same launch sequence and can be
// replayed from a hipGraph):  state = [pos(B) | kvlen(B) | iota(B) | zeros(B) | step | pad(3)]
__global__ void decode_state_init_kernel(int32_t* state, const int32_t* prompt_lens, int B, int step0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        state[i] = prompt_lens[i] + step0;          // position of the token fed next
        state[B + i] = prompt_lens[i] + step0 + 1;  // keys visible to it (itself included)
        state[2 * B + i] = i;
        state[3 * B + i] = 0;
    }
    if (i == 0) state[4 * B] = step0;
}

__global__ void decode_state_advance_kernel(int32_t* state, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { state[i] += 1; state[B + i] += 1; }
    if (i == 0) state[4 * B] += 1;
}

// argmax + append: next_ids[b] = argmax(logits[b]); out_ids[b*ld + step] = next_ids[b]
__global__ __launch_bounds__(1024) void argmax_step_kernel(const float* __restrict__ x, int64_t ld, int64_t* __restrict__ next_ids,
                                                           int64_t* __restrict__ out_ids, int64_t ld_out, const int32_t* __restrict__ step_ptr,
                                                           int N) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const float* r = x + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    auto take = [&](float v, int i) { if (v > best || (v == best && i < idx)) { best = v; idx = i; } };
    if ((N & 3) == 0 && (ld & 3) == 0) {            // 16-byte loads: one row of 32000 logits is 8 loads per thread
        for (int i = threadIdx.x * 4; i < N; i += 4096) {
            const f32x4 v = *(const f32x4*)(r + i);
            take(v[0], i); take(v[1], i + 1); take(v[2], i + 2); take(v[3], i + 3);
        }
    } else {
        for (int i = threadIdx.x; i < N; i += 1024) take(r[i], i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64);
        const int i2 = __shfl_xor(idx, o, 64);
        take(v2, i2);
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) take(bv[w], bi[w]);
        next_ids[blockIdx.x] = idx;
        if (out_ids) out_ids[blockIdx.x * ld_out + (step_ptr ? *step_ptr : 0)] = idx;
    }
}

extern "C" int mc_decode_state_init(int32_t* state, const int32_t* prompt_lens, int B, int step0, void* stream) {
    MC_CHECK_ARG(state && prompt_lens && B > 0, "mc_decode_state_init: bad arguments");
    decode_state_init_kernel<<<(B + 63) / 64, 64, 0, (hipStream_t)stream>>>(state, prompt_lens, B, step0);
    MC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mc_decode_state_advance(int32_t* state, int B, void* stream) {
    decode_state_advance_kernel<<<(B + 63) / 64, 64, 0, (hipStream_t)stream>>>(state, B);
    MC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mc_argmax_step_f32(const void* x, int64_t ld, int64_t* next_ids, int64_t* out_ids, int64_t ld_out,
                                  const int32_t* step_ptr, int M, int N, void* stream) {
    MC_CHECK_ARG(x && next_ids && M > 0 && N > 0, "mc_argmax_step_f32: bad arguments");
    argmax_step_kernel<<<M, 1024, 0, (hipStream_t)stream>>>((const float*)x, ld, next_ids, out_ids, ld_out, step_ptr, N);
    MC_CHECK_LAUNCH();
    return 0;
}
