// Element-wise / reduction kernels of the training step (BASELINE config 5: stage-2 finetune forward + backward).
// They replace autograd through the pieces of MultimodalLlamaDecoderLayer.forward (multimodal_llama.py:408-468) that are not
// GEMMs or attention, the shifted CrossEntropyLoss (:722-733) and the optimiser update; all HBM-bound, 16-byte vector access.
#include "common.h"

// ------------------------------------------------------------------------------------------
// out[c][r] = in[r][c]  (r < R, c < C), columns R..Rp-1 of out zero.  64x64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, int64_t ldi, bf16_t* __restrict__ out, int64_t ldo,
                                                        int R, int C, int Rp) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(int64_t)r * ldi + c] : (bf16_t)0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < Rp) out[(int64_t)c * ldo + r] = tile[tx][i];
    }
}

extern "C" int mc_transpose_bf16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, int Rp, void* stream) {
    MC_CHECK_ARG(in && out && R > 0 && C > 0 && Rp >= R, "mc_transpose_bf16: bad arguments");
    dim3 grid((C + 63) / 64, (Rp + 63) / 64);
    transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)in, ldi, (bf16_t*)out, ldo, R, C, Rp);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// LocalLoRA routing of the low-rank activations: t is [M, n_cols] = one r-wide block per (linear, adapter), adapters fastest
// (n_cols = n_linears * n_adapters * r); row m keeps the blocks of its adapter (row_adapter[m]) and is zeroed elsewhere — the
// per-token mask-sum of multimodal_llama.py:262-268 applied where it is cheap.
__global__ __launch_bounds__(256) void lora_mask_kernel(bf16_t* __restrict__ t, int64_t ld, const int32_t* __restrict__ row_adapter, int M, int r,
                                                        int n_adapters, int n_cols) {
    const int nv = n_cols >> 3;
    const int64_t total = (int64_t)M * nv;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / nv), cv = (int)(i % nv);
        if (((cv * 8) / r) % n_adapters != row_adapter[m]) *(u32x4*)(t + (int64_t)m * ld + cv * 8) = (u32x4){0u, 0u, 0u, 0u};
    }
}

extern "C" int mc_lora_mask_rows_bf16(void* t, int64_t ld, const int32_t* row_adapter, int M, int r, int n_adapters, int n_cols, void* stream) {
    MC_CHECK_ARG(t && row_adapter && M > 0 && r > 0 && r % 8 == 0 && n_adapters > 0 && ld % 8 == 0 && n_cols > 0 && n_cols % (n_adapters * r) == 0,
                 "mc_lora_mask_rows_bf16: bad arguments");
    const int64_t total = (int64_t)M * (n_cols >> 3);
    lora_mask_kernel<<<(int)min((int64_t)8192, (total + 255) / 256), 256, 0, (hipStream_t)stream>>>((bf16_t*)t, ld, row_adapter, M, r, n_adapters, n_cols);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward (the Q-Former projector's BertLayer norms, multimodal_projector/Qformer.py:112-130, trained in the audio stage-2
// recipe): y = xhat * g + b, xhat = (x - mean) * rstd.
//   dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat));   t = dy * xhat  (dgamma = column sums of t, dbeta = column sums of dy)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ g,
                                                            const bf16_t* __restrict__ dy, int64_t ldy, bf16_t* __restrict__ dx, int64_t ldd,
                                                            bf16_t* __restrict__ t_out, int64_t ldt, int D, float eps) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const bf16_t* xr = x + (int64_t)row * ldx;
    const bf16_t* dyr = dy + (int64_t)row * ldy;
    float s1 = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) s1 += (float)xr[i];
    const float mean = block_sum(s1, red) / D;
    float sv = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) { const float d = (float)xr[i] - mean; sv += d * d; }
    const float rstd = rsqrtf(block_sum(sv, red) / D + eps);
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) {
        const float xh = ((float)xr[i] - mean) * rstd, dg = (float)dyr[i] * (float)g[i];
        a += dg; b += dg * xh;
    }
    a = block_sum(a, red) / D;
    b = block_sum(b, red) / D;
    for (int i = threadIdx.x; i < D; i += 256) {
        const float xh = ((float)xr[i] - mean) * rstd, dyv = (float)dyr[i];
        dx[(int64_t)row * ldd + i] = (bf16_t)(rstd * (dyv * (float)g[i] - a - xh * b));
        if (t_out) t_out[(int64_t)row * ldt + i] = (bf16_t)(dyv * xh);
    }
}

extern "C" int mc_layernorm_bwd_bf16(const void* x, int64_t ldx, const void* g, const void* dy, int64_t ldy, void* dx, int64_t ldd, void* t_out,
                                     int64_t ldt, int M, int D, float eps, void* stream) {
    MC_CHECK_ARG(x && g && dy && dx && M > 0 && D > 0, "mc_layernorm_bwd_bf16: bad arguments");
    layernorm_bwd_kernel<<<M, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (const bf16_t*)g, (const bf16_t*)dy, ldy, (bf16_t*)dx, ldd,
                                                             (bf16_t*)t_out, ldt, D, eps);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// nn.Dropout on the LoRA input (LocalLoraLinear.forward, multimodal_llama.py:133-148: lora_A(lora_dropout(x)), p = 0.05 in the stage-2
// scripts).  Counter-based mask: element e = m*K + k of stream `stream_id` is kept iff word (e & 3) of
// Philox4x32-10(counter = (e >> 2 lo, e >> 2 hi, stream_id, 0), key = (seed_lo, seed_hi)) >= p * 2^32, so the backward pass (and the
// CPU oracle, oracle/philox.py) regenerates the identical mask from (seed, stream_id) instead of storing it.
//   out[m][k] = (accumulate ? out[m][k] : 0) + alpha * x[m][k] * keep / (1 - p)
__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ out, int64_t ldo, int M, int K,
                                                      uint32_t thr, float inv_keep, uint32_t seed_lo, uint32_t seed_hi, uint32_t stream_id,
                                                      int accumulate, float alpha) {
    const int nv = K >> 3;
    const int64_t total = (int64_t)M * nv;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / nv), c = (int)(i % nv);
        const uint64_t e4 = ((uint64_t)m * (uint64_t)K + (uint64_t)c * 8) >> 2;          // K % 8 == 0: a vector never straddles two counters' quads
        uint32_t r0[4], r1[4];
        philox4((uint32_t)e4, (uint32_t)(e4 >> 32), stream_id, 0u, seed_lo, seed_hi, r0);
        philox4((uint32_t)(e4 + 1), (uint32_t)((e4 + 1) >> 32), stream_id, 0u, seed_lo, seed_hi, r1);
        const bf16x8 xv = *(const bf16x8*)(x + (int64_t)m * ldx + c * 8);
        bf16x8 o;
        if (accumulate) o = *(const bf16x8*)(out + (int64_t)m * ldo + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t r = j < 4 ? r0[j] : r1[j - 4];
            const float v = r >= thr ? alpha * (float)xv[j] * inv_keep : 0.f;
            o[j] = (bf16_t)(accumulate ? (float)o[j] + v : v);
        }
        *(bf16x8*)(out + (int64_t)m * ldo + c * 8) = o;
    }
}

extern "C" int mc_dropout_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int M, int K, float p, unsigned long long seed,
                               unsigned int stream_id, int accumulate, float alpha, void* stream) {
    MC_CHECK_ARG(x && out && M > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "mc_dropout_bf16: bad arguments");
    MC_CHECK_ARG(p >= 0.f && p < 1.f, "mc_dropout_bf16: dropout probability has to be between 0 and 1, but got %g", (double)p);
    const double t = (double)p * 4294967296.0;
    const uint32_t thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
    const int64_t total = (int64_t)M * (K >> 3);
    dropout_kernel<<<(int)min((int64_t)8192, (total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, ldx, (bf16_t*)out, ldo, M, K, thr, 1.0f / (1.0f - p), (uint32_t)seed, (uint32_t)(seed >> 32), stream_id, accumulate, alpha);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// RMSNorm backward (LlamaRMSNorm: y = g * x * rs, rs = rsqrt(mean(x^2) + eps)):
//   dx = rs * (g*dy) - x * rs^3 * mean(x * g*dy)   (+ dres: the residual branch's gradient)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ g,
                                                          const bf16_t* __restrict__ dy, int64_t ldy, const bf16_t* __restrict__ dres,
                                                          int64_t ldr, bf16_t* __restrict__ dx, int64_t ldd, int D, float eps) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const bf16_t* xr = x + (int64_t)row * ldx;
    const bf16_t* dyr = dy + (int64_t)row * ldy;
    const int nv = D >> 3;
    float s2 = 0.f, sxg = 0.f;
    for (int i = threadIdx.x; i < nv; i += 256) {
        const bf16x8 xv = *(const bf16x8*)(xr + i * 8), gv = *(const bf16x8*)(g + i * 8), dv = *(const bf16x8*)(dyr + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xf = (float)xv[j];
            s2 += xf * xf;
            sxg += xf * (float)gv[j] * (float)dv[j];
        }
    }
    s2 = block_sum(s2, red);
    sxg = block_sum(sxg, red);
    const float rs = rsqrtf(s2 / D + eps);
    const float coef = rs * rs * rs * sxg / D;
    for (int i = threadIdx.x; i < nv; i += 256) {
        const bf16x8 xv = *(const bf16x8*)(xr + i * 8), gv = *(const bf16x8*)(g + i * 8), dv = *(const bf16x8*)(dyr + i * 8);
        bf16x8 rv;
        if (dres) rv = *(const bf16x8*)(dres + (int64_t)row * ldr + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = rs * (float)gv[j] * (float)dv[j] - (float)xv[j] * coef;
            if (dres) v += (float)rv[j];
            o[j] = (bf16_t)v;
        }
        *(bf16x8*)(dx + (int64_t)row * ldd + i * 8) = o;
    }
}

extern "C" int mc_rmsnorm_bwd_bf16(const void* x, int64_t ldx, const void* g, const void* dy, int64_t ldy, const void* dres, int64_t ldr,
                                   void* dx, int64_t ldd, int M, int D, float eps, void* stream) {
    MC_CHECK_ARG(x && g && dy && dx && M > 0 && D > 0 && D % 8 == 0 && (ldx | ldy | ldd | ldr) % 8 == 0, "mc_rmsnorm_bwd_bf16: bad arguments");
    rmsnorm_bwd_kernel<<<M, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (const bf16_t*)g, (const bf16_t*)dy, ldy, (const bf16_t*)dres,
                                                           ldr, (bf16_t*)dx, ldd, D, eps);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// SwiGLU backward: inter = silu(gate) * up  ->  dgate = dinter * up * silu'(gate), dup = dinter * silu(gate)
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, int64_t ld, const bf16_t* __restrict__ dinter, int64_t ldi,
                                                         bf16_t* __restrict__ dgu, int64_t ldg, int M, int I) {
    const int nv = I >> 3;
    const int64_t total = (int64_t)M * nv;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / nv), c = (int)(i % nv);
        const bf16x8 g = *(const bf16x8*)(gu + (int64_t)m * ld + c * 8), u = *(const bf16x8*)(gu + (int64_t)m * ld + I + c * 8);
        const bf16x8 d = *(const bf16x8*)(dinter + (int64_t)m * ldi + c * 8);
        bf16x8 og, ou;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gf = (float)g[j], sg = mc_sigmoid(gf), df = (float)d[j];
            ou[j] = (bf16_t)(df * gf * sg);
            og[j] = (bf16_t)(df * (float)u[j] * sg * (1.0f + gf * (1.0f - sg)));
        }
        *(bf16x8*)(dgu + (int64_t)m * ldg + c * 8) = og;
        *(bf16x8*)(dgu + (int64_t)m * ldg + I + c * 8) = ou;
    }
}

extern "C" int mc_swiglu_bwd_bf16(const void* gate_up, int64_t ld, const void* dinter, int64_t ldi, void* dgate_up, int64_t ldg, int M, int I,
                                  void* stream) {
    MC_CHECK_ARG(gate_up && dinter && dgate_up && M > 0 && I > 0 && I % 8 == 0 && (ld | ldi | ldg) % 8 == 0, "mc_swiglu_bwd_bf16: bad arguments");
    const int64_t total = (int64_t)M * (I >> 3);
    swiglu_bwd_kernel<<<(int)min((int64_t)8192, (total + 255) / 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)gate_up, ld, (const bf16_t*)dinter,
                                                                                                    ldi, (bf16_t*)dgate_up, ldg, M, I);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// activation forward / backward on a saved pre-activation (mm_projector's nn.GELU, multimodal_projector/builder.py:208-215)
__device__ __forceinline__ float act_grad(float x, int act) {
    switch (act) {
        case MC_ACT_GELU: {
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
            return cdf + x * 0.3989422804014327f * __expf(-0.5f * x * x);
        }
        case MC_ACT_QUICK_GELU: { const float s = mc_sigmoid(1.702f * x); return s * (1.0f + 1.702f * x * (1.0f - s)); }
        case MC_ACT_SILU: { const float s = mc_sigmoid(x); return s * (1.0f + x * (1.0f - s)); }
        case MC_ACT_RELU: return x > 0.f ? 1.0f : 0.f;
        default: return 1.0f;
    }
}

// dy == null: out = act(pre);  else out = dy * act'(pre)
__global__ __launch_bounds__(256) void act_kernel(const bf16_t* __restrict__ pre, const bf16_t* __restrict__ dy, bf16_t* __restrict__ out, int64_t n8,
                                                  int act) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const bf16x8 p = *(const bf16x8*)(pre + i * 8);
        bf16x8 o;
        if (dy) {
            const bf16x8 d = *(const bf16x8*)(dy + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)d[j] * act_grad((float)p[j], act));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16_t)mc_act((float)p[j], act);
        }
        *(bf16x8*)(out + i * 8) = o;
    }
}

extern "C" int mc_act_bf16(const void* pre, const void* dy, void* out, int64_t n, int act, void* stream) {
    MC_CHECK_ARG(pre && out && n > 0 && n % 8 == 0, "mc_act_bf16: n must be a positive multiple of 8");
    act_kernel<<<(int)min((int64_t)8192, (n / 8 + 255) / 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)pre, (const bf16_t*)dy, (bf16_t*)out, n / 8, act);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// shifted cross-entropy (multimodal_llama.py:722-733): per row m with target label[m] (already shifted; < 0 = ignored)
//   loss_rows[m] = logsumexp(logits[m]) - logits[m][label];  dlogits[m] = (softmax - onehot) * inv_n   (bf16, zero for ignored rows)
__global__ __launch_bounds__(256) void ce_loss_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                      float* __restrict__ loss_rows, bf16_t* __restrict__ dlogits, int64_t ldd, int V, float inv_n) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const float* lr = logits + (int64_t)row * ld;
    bf16_t* dr = dlogits ? dlogits + (int64_t)row * ldd : nullptr;         // dlogits == NULL: loss only (forward(labels=...))
    const int64_t lab = labels[row];
    const int nv = V >> 2;
    if (lab < 0) {
        loss_rows[row] = 0.f;
        if (dr)
            for (int i = threadIdx.x; i < nv; i += 256) *(bf16x4*)(dr + i * 4) = (bf16x4){(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        return;
    }
    float mx = -3.0e38f;
    for (int i = threadIdx.x; i < nv; i += 256) {
        const f32x4 v = *(const f32x4*)(lr + i * 4);
        mx = fmaxf(mx, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
    }
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float se = 0.f;
    for (int i = threadIdx.x; i < nv; i += 256) {
        const f32x4 v = *(const f32x4*)(lr + i * 4);
        se += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx);
    }
    se = block_sum(se, red);
    const float inv = 1.0f / se;
    if (threadIdx.x == 0) loss_rows[row] = logf(se) + mx - lr[lab];
    if (!dr) return;
    for (int i = threadIdx.x; i < nv; i += 256) {
        const f32x4 v = *(const f32x4*)(lr + i * 4);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = __expf(v[j] - mx) * inv;
            if (i * 4 + j == lab) pj -= 1.0f;
            o[j] = (bf16_t)(pj * inv_n);
        }
        *(bf16x4*)(dr + i * 4) = o;
    }
}

extern "C" int mc_ce_loss_f32(const float* logits, int64_t ld, const int64_t* labels, float* loss_rows, void* dlogits_bf16, int64_t ldd, int M,
                              int V, float inv_n, void* stream) {
    MC_CHECK_ARG(logits && labels && loss_rows && M > 0 && V > 0 && V % 4 == 0 && ld % 4 == 0 && (!dlogits_bf16 || ldd % 4 == 0), "mc_ce_loss_f32: bad arguments");
    ce_loss_kernel<<<M, 256, 0, (hipStream_t)stream>>>(logits, ld, labels, loss_rows, (bf16_t*)dlogits_bf16, ldd, V, inv_n);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// out[c] = sum_m x[m][c]   (bias gradients, prefix/suffix token gradients); fp32 result.  One workgroup per 64 columns.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int64_t ld, float* __restrict__ out, int M, int C) {
    __shared__ float part[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    float s = 0.f;
    if (col < C)
        for (int m = ty; m < M; m += 4) s += (float)x[(int64_t)m * ld + col];
    part[ty][threadIdx.x & 63] = s;
    __syncthreads();
    if (ty == 0 && col < C) out[col] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

extern "C" int mc_colsum_bf16(const void* x, int64_t ld, float* out, int M, int C, void* stream) {
    MC_CHECK_ARG(x && out && M > 0 && C > 0, "mc_colsum_bf16: bad arguments");
    colsum_kernel<<<(C + 63) / 64, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ld, out, M, C);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// in-place rotate-half RoPE on the first n_heads heads of every row (transformers 4.31 apply_rotary_pos_emb via
// multimodal_llama.py:281-282); sign = -1 applies the inverse rotation = the backward of the forward rotation.
__global__ __launch_bounds__(256) void rope_inplace_kernel(bf16_t* __restrict__ x, int64_t ld, const int32_t* __restrict__ row_pos,
                                                           const float* __restrict__ cosT, const float* __restrict__ sinT, int n_heads, int D,
                                                           float sign) {
    const int r = blockIdx.x;
    const int pos = row_pos[r];
    const int half = D >> 1, cpd = half >> 3;
    bf16_t* xr = x + (int64_t)r * ld;
    const float* cr = cosT + (int64_t)pos * half;
    const float* sr = sinT + (int64_t)pos * half;
    for (int it = threadIdx.x; it < n_heads * cpd; it += 256) {
        const int hh = it / cpd, ch = it % cpd;
        bf16_t* s0 = xr + hh * D + ch * 8;
        const bf16x8 x1 = *(const bf16x8*)s0, x2 = *(const bf16x8*)(s0 + half);
        bf16x8 o1, o2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float c = cr[ch * 8 + j], s = sign * sr[ch * 8 + j];
            const float a = (float)x1[j], b = (float)x2[j];
            o1[j] = (bf16_t)(a * c - b * s);
            o2[j] = (bf16_t)(b * c + a * s);
        }
        *(bf16x8*)s0 = o1;
        *(bf16x8*)(s0 + half) = o2;
    }
}

extern "C" int mc_rope_inplace_bf16(void* x, int64_t ld, const int32_t* row_pos, const float* cos_table, const float* sin_table, int M,
                                    int n_heads, int D, float sign, void* stream) {
    MC_CHECK_ARG(x && row_pos && cos_table && sin_table && M > 0 && n_heads > 0 && D % 16 == 0 && ld % 8 == 0, "mc_rope_inplace_bf16: bad arguments");
    rope_inplace_kernel<<<M, 256, 0, (hipStream_t)stream>>>((bf16_t*)x, ld, row_pos, cos_table, sin_table, n_heads, D, sign);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW semantics, the optimiser HF Trainer builds in llava_trainer.py:210-288) on fp32 master weights,
// refreshing the bf16 working copy in the same pass.  grad is scaled by grad_scale first (1/world_size after the all-reduce sum).
__device__ __forceinline__ float adamw_one(float& pi, float gi, float& mi, float& vi, float lr, float b1, float b2, float eps, float wd, float bc1,
                                           float bc2, float grad_scale) {
    gi *= grad_scale;
    pi *= (1.0f - lr * wd);
    mi = b1 * mi + (1.0f - b1) * gi;
    vi = b2 * vi + (1.0f - b2) * gi * gi;
    pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    return pi;
}

// 16-byte accesses on the four fp32 streams (30 bytes move per parameter: the kernel is a pure HBM stream); n4 = n / 4 vectors, the
// tail (n % 4 elements) is handled by the last threads one element at a time
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ p16, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2, float grad_scale) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 pv = ((f32x4*)p)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        const f32x4 gv = ((const f32x4*)g)[i];
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = pv[j], mj = mv[j], vj = vv[j];
            o[j] = (bf16_t)adamw_one(pj, gv[j], mj, vj, lr, b1, b2, eps, wd, bc1, bc2, grad_scale);
            pv[j] = pj; mv[j] = mj; vv[j] = vj;
        }
        ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
        if (p16) ((bf16x4*)p16)[i] = o;
    }
    const int64_t t = (n4 << 2) + blockIdx.x * 256LL + threadIdx.x;
    if (t < n) {
        float pj = p[t], mj = m[t], vj = v[t];
        const float r = adamw_one(pj, g[t], mj, vj, lr, b1, b2, eps, wd, bc1, bc2, grad_scale);
        p[t] = pj; m[t] = mj; v[t] = vj;
        if (p16) p16[t] = (bf16_t)r;
    }
}

extern "C" int mc_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    MC_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "mc_adamw_f32: bad arguments");
    MC_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0 && (uintptr_t)param_bf16 % 8 == 0,
                 "mc_adamw_f32: buffers must be 16-byte aligned (bf16 copy: 8)");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    adamw_kernel<<<(int)min((int64_t)8192, (n / 4 + 256) / 256), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, (bf16_t*)param_bf16, n, lr,
                                                                                            beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
    MC_CHECK_LAUNCH();
    return 0;
}

// AdamW over a flat buffer whose tensors belong to TWO learning-rate groups chosen per element (llava_trainer.py:210-290: with
// --mm_projector_lr / --mm_language_lr the `lora_A.default` / `lora_B.default` tensors take the projector group's rate, the modal adapters
// the base rate; here an adapter's A rows / B columns are slices of stacked tensors).  The buffer is described by chunks
// {offset, length, index of the chunk's first element inside its tensor, period, width}: element e of a tensor is in the ALTERNATE group
// when e mod period < width (A_in [n_lin * nA * r, K]: period nA*r*K, width r*K; B_cat [N, nA*r]: period nA*r, width r).  Offsets,
// lengths, periods and widths are multiples of 4, so a 16-byte vector never straddles two groups.
struct AdamwSeg { long long off; int n; int idx0; int period; int width; };
static_assert(sizeof(AdamwSeg) == sizeof(mc_adamw_seg), "mc_adamw_seg layout");

__global__ __launch_bounds__(256) void adamw_seg_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                        bf16_t* __restrict__ p16, const AdamwSeg* __restrict__ segs, int n_segs, float lr, float lr_alt,
                                                        float b1, float b2, float eps, float wd, float bc1, float bc2, float grad_scale) {
    for (int sidx = blockIdx.x; sidx < n_segs; sidx += gridDim.x) {
        const AdamwSeg sg = segs[sidx];
        const int n4 = sg.n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            const int64_t e = (sg.off >> 2) + i;
            const bool alt = sg.period > 0 && (uint32_t)(sg.idx0 + 4 * i) % (uint32_t)sg.period < (uint32_t)sg.width;
            const float l = alt ? lr_alt : lr;
            f32x4 pv = ((f32x4*)p)[e], mv = ((f32x4*)m)[e], vv = ((f32x4*)v)[e];
            const f32x4 gv = ((const f32x4*)g)[e];
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = pv[j], mj = mv[j], vj = vv[j];
                o[j] = (bf16_t)adamw_one(pj, gv[j], mj, vj, l, b1, b2, eps, wd, bc1, bc2, grad_scale);
                pv[j] = pj; mv[j] = mj; vv[j] = vj;
            }
            ((f32x4*)p)[e] = pv; ((f32x4*)m)[e] = mv; ((f32x4*)v)[e] = vv;
            if (p16) ((bf16x4*)p16)[e] = o;
        }
    }
}

extern "C" int mc_adamw_segments_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16, const mc_adamw_seg* segs_dev,
                                     int n_segs, float lr, float lr_alt, float beta1, float beta2, float eps, float weight_decay, int step,
                                     float grad_scale, void* stream) {
    MC_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && segs_dev && n_segs > 0 && step >= 1, "mc_adamw_segments_f32: bad arguments");
    MC_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0 && (uintptr_t)param_bf16 % 8 == 0,
                 "mc_adamw_segments_f32: buffers must be 16-byte aligned (bf16 copy: 8)");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    adamw_seg_kernel<<<min(n_segs, 16384), 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, (bf16_t*)param_bf16, (const AdamwSeg*)segs_dev,
                                                                         n_segs, lr, lr_alt, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
    MC_CHECK_LAUNCH();
    return 0;
}

// y += alpha * x over flat fp32 buffers (gradient accumulation over micro-batches: run_finetune_vision_damc.sh:45 --gradient_accumulation_steps 4)
__global__ __launch_bounds__(256) void axpy_f32_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, float alpha) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 a = ((f32x4*)y)[i];
        const f32x4 b = ((const f32x4*)x)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = fmaf(alpha, b[j], a[j]);
        ((f32x4*)y)[i] = a;
    }
    const int64_t t = (n4 << 2) + blockIdx.x * 256LL + threadIdx.x;
    if (t < n) y[t] = fmaf(alpha, x[t], y[t]);
}

extern "C" int mc_axpy_f32(float* y, const float* x, int64_t n, float alpha, void* stream) {
    MC_CHECK_ARG(y && x && n > 0 && ((uintptr_t)y | (uintptr_t)x) % 16 == 0, "mc_axpy_f32: bad arguments");
    axpy_f32_kernel<<<(int)min((int64_t)8192, (n / 4 + 256) / 256), 256, 0, (hipStream_t)stream>>>(y, x, n, alpha);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// fp32 -> bf16 cast of a flat buffer (working copies of the trainable parameters)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = (bf16_t)x[i];
}

extern "C" int mc_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream) {
    MC_CHECK_ARG(x && y && n > 0, "mc_cast_f32_bf16: bad arguments");
    cast_f32_bf16_kernel<<<(int)min((int64_t)8192, (n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, (bf16_t*)y, n);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Weight-gradient GEMM in "TN" form:  out[p][q] = alpha * sum_m a[m][p] * b[m][q]   (fp32 out; a [M, P], b [M, Q] row-major bf16).
// The LoRA gradients of the finetune step are of this shape (dB = dy^T T, dA = mask(dT)^T x: train/step.py): the reduction runs over the
// token rows, the slow dimension of both operands.  Instead of transposing the activations in HBM (two full passes) the 64-row chunks are
// staged row-major in LDS and both MFMA operands are fetched with the transposing LDS read (ds_read_b64_tr_b16), which hands lane i of a
// 16-lane group column i of a 4-row block.  Rows are padded to 160 B so that the 8 rows one read touches start on distinct 8-bank windows.
// The work is HBM / L2-bound (2*M*P*Q flops on M*(P+Q)*2 bytes with Q or P = n_adapters*r = 256): 64x64 output tiles, one per workgroup;
// when that gives fewer than ~2 workgroups per CU the token range is split in `splits` slabs reduced in fixed order by a second kernel.
// Up to 3 problems of the same shape share a launch (the q / k / v projections).
#define TN_ROWB 160
struct TnParams {
    const bf16_t* a[3]; const bf16_t* b[3]; float* out[3];
    int64_t lda, ldb, ldo;
    int M, P, Q, splits;
    float alpha;
    float* slabs;          // [problem][split][P][Q] when splits > 1
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(TnParams p) {
    __shared__ __attribute__((aligned(16))) char la[64 * TN_ROWB];
    __shared__ __attribute__((aligned(16))) char lb[64 * TN_ROWB];
    const int prob = blockIdx.z / p.splits, split = blockIdx.z % p.splits;
    // (selects, not a dynamically indexed kernel-argument array: that would be re-fetched from memory inside the loop)
    const bf16_t* A = prob == 0 ? p.a[0] : prob == 1 ? p.a[1] : p.a[2];
    const bf16_t* B = prob == 0 ? p.b[0] : prob == 1 ? p.b[1] : p.b[2];
    const int p0 = blockIdx.y * 64, q0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1;
    const int g = lane >> 4, c16 = lane & 15, tq = c16 >> 2, tp = c16 & 3;
    // rows of this split, in 64-row chunks
    const int rows_per = ((p.M + p.splits - 1) / p.splits + 63) / 64 * 64;
    const int m_begin = split * rows_per, m_end = min(p.M, m_begin + rows_per);
    // staging: thread t loads 16 B at (row t>>3 [+32], col chunk t&7) of each operand
    const int srow = tid >> 3, sch = tid & 7;
    u32x4 ra[2], rb[2];
    auto load = [&](int mb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // unconditional loads from clamped (always valid) addresses, zeroed afterwards: no divergent branch around a load
            const int m = mb + srow + 32 * h;
            const bool ok = m < m_end;
            const int mc = min(m, p.M - 1);
            const int pc = p0 + sch * 8, qc = q0 + sch * 8;
            const u32x4 va = *(const u32x4*)(A + (int64_t)mc * p.lda + (pc < p.P ? pc : 0));
            const u32x4 vb = *(const u32x4*)(B + (int64_t)mc * p.ldb + (qc < p.Q ? qc : 0));
            ra[h] = (ok && pc < p.P) ? va : (u32x4){0u, 0u, 0u, 0u};
            rb[h] = (ok && qc < p.Q) ? vb : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (m_begin < m_end) load(m_begin);
    for (int mb = m_begin; mb < m_end; mb += 64) {
        __syncthreads();                                   // previous chunk's fragment reads are done
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *(u32x4*)(la + (srow + 32 * h) * TN_ROWB + sch * 16) = ra[h];
            *(u32x4*)(lb + (srow + 32 * h) * TN_ROWB + sch * 16) = rb[h];
        }
        __syncthreads();
        if (mb + 64 < m_end) load(mb + 64);                // next chunk's global loads fly under this chunk's MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r_lo = ks * 32 + g * 4 + tq, r_hi = r_lo + 16;
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int colb = (wp * 32 + i * 16 + tp * 4) * 2;
                const bf16x4 lo = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(la + r_lo * TN_ROWB + colb));
                const bf16x4 hi = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(la + r_hi * TN_ROWB + colb));
                fa[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const int colq = (wq * 32 + i * 16 + tp * 4) * 2;
                const bf16x4 lo2 = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(lb + r_lo * TN_ROWB + colq));
                const bf16x4 hi2 = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(lb + r_hi * TN_ROWB + colq));
                fb[i] = (bf16x8){lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mc_mfma_16x16x32(fa[i], fb[j], acc[i][j]);
        }
    }
    // D[row = 4g + r][col = c16] of the (i, j) block: out[p0 + wp*32 + i*16 + 4g + r][q0 + wq*32 + j*16 + c16]
    float* dst = p.splits > 1 ? p.slabs + ((int64_t)(prob * p.splits + split) * p.P) * p.Q : p.out[prob];
    const int64_t ldd = p.splits > 1 ? p.Q : p.ldo;
    const float sc = p.splits > 1 ? 1.0f : p.alpha;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = p0 + wp * 32 + i * 16 + g * 4 + r, qc = q0 + wq * 32 + j * 16 + c16;
                if (pr < p.P && qc < p.Q) dst[(int64_t)pr * ldd + qc] = acc[i][j][r] * sc;
            }
}

// Wide variant: the workgroup tile spans 256 columns of one operand and 64 of the other (TPB x TQB 64-column blocks = 1x4 or 4x1), so when
// P or Q is the rank dimension (n_adapters * r = 256 per linear) the large activation is streamed exactly once.  Each wave owns a 64x64
// output block (4x4 MFMA tiles: 16 MFMAs per 16 transposing reads per 32-row k-step); the next 64-row chunk (40 KiB) is prefetched into
// registers under the MFMAs of the current one.
template <int TPB, int TQB>
__global__ __launch_bounds__(256) void gemm_tn_wide_kernel(TnParams p) {
    constexpr int TP = TPB * 64, TQ = TQB * 64;
    constexpr int SA = TP * 2 + 32, SB = TQ * 2 + 32;          // padded row strides (bytes): 8 consecutive rows start on distinct 8-bank windows
    constexpr int CPR = (TP + TQ) / 8;                          // 16-byte chunks per staged row (A part then B part)
    constexpr int NLD = 64 * CPR / 256;                         // chunks per thread per 64-row step
    __shared__ __attribute__((aligned(16))) char lds[64 * SA + 64 * SB];
    char* const la = lds;
    char* const lb = lds + 64 * SA;
    const int prob = blockIdx.z / p.splits, split = blockIdx.z % p.splits;
    const bf16_t* A = prob == 0 ? p.a[0] : prob == 1 ? p.a[1] : p.a[2];
    const bf16_t* B = prob == 0 ? p.b[0] : prob == 1 ? p.b[1] : p.b[2];
    const int p0 = blockIdx.y * TP, q0 = blockIdx.x * TQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pa0 = TPB == 1 ? 0 : wave * 64, qb0 = TPB == 1 ? wave * 64 : 0;
    const int g = lane >> 4, c16 = lane & 15, tq = c16 >> 2, tp = c16 & 3;
    const int rows_per = ((p.M + p.splits - 1) / p.splits + 63) / 64 * 64;
    const int m_begin = split * rows_per, m_end = min(p.M, m_begin + rows_per);
    // staging plan (loop invariant): chunk idx = tid + 256 i  ->  row idx / CPR, column chunk idx % CPR.  The A / B choice is folded into
    // one base pointer + row stride per chunk; loads are unconditional from clamped addresses and zeroed afterwards.
    int srow[NLD], lds_off[NLD];
    const bf16_t* gsrc[NLD];
    int64_t gld[NLD];
    bool inb[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / CPR, c = idx - r * CPR;
        const bool isA = c < TP / 8;
        srow[i] = r;
        const int col = isA ? p0 + c * 8 : q0 + (c - TP / 8) * 8;
        inb[i] = isA ? col < p.P : col < p.Q;
        gld[i] = isA ? p.lda : p.ldb;
        gsrc[i] = (isA ? A : B) + (inb[i] ? col : 0);
        lds_off[i] = isA ? r * SA + c * 16 : 64 * SA + r * SB + (c - TP / 8) * 16;
    }
    u32x4 reg[NLD];
    auto load = [&](int mb) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int m = mb + srow[i];
            const u32x4 v = *(const u32x4*)(gsrc[i] + (int64_t)min(m, p.M - 1) * gld[i]);
            reg[i] = (inb[i] && m < m_end) ? v : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (m_begin < m_end) load(m_begin);
    for (int mb = m_begin; mb < m_end; mb += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NLD; ++i) *(u32x4*)(lds + lds_off[i]) = reg[i];
        __syncthreads();
        if (mb + 64 < m_end) load(mb + 64);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r_lo = ks * 32 + g * 4 + tq, r_hi = r_lo + 16;
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ca = (pa0 + i * 16 + tp * 4) * 2, cb = (qb0 + i * 16 + tp * 4) * 2;
                const bf16x4 lo = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(la + r_lo * SA + ca));
                const bf16x4 hi = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(la + r_hi * SA + ca));
                fa[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const bf16x4 lo2 = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(lb + r_lo * SB + cb));
                const bf16x4 hi2 = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(lb + r_hi * SB + cb));
                fb[i] = (bf16x8){lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mc_mfma_16x16x32(fa[i], fb[j], acc[i][j]);
        }
    }
    float* dst = p.splits > 1 ? p.slabs + ((int64_t)(prob * p.splits + split) * p.P) * p.Q : p.out[prob];
    const int64_t ldd = p.splits > 1 ? p.Q : p.ldo;
    const float sc = p.splits > 1 ? 1.0f : p.alpha;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = p0 + pa0 + i * 16 + g * 4 + r, qc = q0 + qb0 + j * 16 + c16;
                if (pr < p.P && qc < p.Q) dst[(int64_t)pr * ldd + qc] = acc[i][j][r] * sc;
            }
}

__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(TnParams p) {
    const int prob = blockIdx.y;
    const int64_t n = (int64_t)p.P * p.Q;
    const float* s = p.slabs + (int64_t)prob * p.splits * n;
    float* out = p.out[prob];
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = s[i];
        for (int k = 1; k < p.splits; ++k) v += s[k * n + i];
        out[(i / p.Q) * p.ldo + (i % p.Q)] = v * p.alpha;
    }
}

// n_problems (1..3) products of the same shape in one launch.  workspace: n_problems * splits * P * Q floats when the kernel decides to
// split (query the size with mc_gemm_tn_workspace_floats); may be NULL when that returns 0.
// variant: 1 = 64x256 tile (Q spans the rank), 2 = 256x64 (P spans the rank), 0 = 64x64
static int tn_variant(int P, int Q) {
    if (Q % 256 == 0 && Q <= 512 && P >= Q) return 1;
    if (P % 256 == 0 && P <= 1024 && Q >= P) return 2;
    return 0;
}
static int tn_splits(int M, int P, int Q, int n_problems) {
    const int v = tn_variant(P, Q);
    const int tp = v == 2 ? 256 : 64, tq = v == 1 ? 256 : 64;
    const int64_t tiles = (int64_t)((P + tp - 1) / tp) * ((Q + tq - 1) / tq) * n_problems;
    const int64_t want = v ? 256 : 512;
    int s = 1;
    while (tiles * s < want && s < 8 && M / (s * 2) >= 256) s *= 2;
    return s;
}
extern "C" int mc_gemm_tn_workspace_floats(int M, int P, int Q, int n_problems, int64_t* floats) {
    MC_CHECK_ARG(floats && M > 0 && P > 0 && Q > 0 && n_problems >= 1 && n_problems <= 3, "mc_gemm_tn_workspace_floats: bad arguments");
    const int s = tn_splits(M, P, Q, n_problems);
    *floats = s > 1 ? (int64_t)n_problems * s * P * Q : 0;
    return 0;
}
extern "C" int mc_gemm_tn_bf16(const void* const* a, int64_t lda, const void* const* b, int64_t ldb, float* const* out, int64_t ldo, int n_problems,
                               int M, int P, int Q, float alpha, float* workspace, void* stream) {
    MC_CHECK_ARG(a && b && out && n_problems >= 1 && n_problems <= 3 && M > 0 && P > 0 && Q > 0, "mc_gemm_tn_bf16: bad arguments");
    MC_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && P % 8 == 0 && Q % 8 == 0, "mc_gemm_tn_bf16: P, Q and the row strides must be multiples of 8");
    TnParams p{};
    for (int i = 0; i < n_problems; ++i) {
        MC_CHECK_ARG(a[i] && b[i] && out[i] && ((uintptr_t)a[i] % 16) == 0 && ((uintptr_t)b[i] % 16) == 0, "mc_gemm_tn_bf16: null / unaligned operand %d", i);
        p.a[i] = (const bf16_t*)a[i]; p.b[i] = (const bf16_t*)b[i]; p.out[i] = out[i];
    }
    p.lda = lda; p.ldb = ldb; p.ldo = ldo; p.M = M; p.P = P; p.Q = Q; p.alpha = alpha;
    p.splits = tn_splits(M, P, Q, n_problems);
    p.slabs = workspace;
    MC_CHECK_ARG(p.splits == 1 || workspace, "mc_gemm_tn_bf16: workspace missing (%d splits)", p.splits);
    hipStream_t s = (hipStream_t)stream;
    const int variant = tn_variant(P, Q);
    if (variant == 1) gemm_tn_wide_kernel<1, 4><<<dim3((Q + 255) / 256, (P + 63) / 64, n_problems * p.splits), 256, 0, s>>>(p);
    else if (variant == 2) gemm_tn_wide_kernel<4, 1><<<dim3((Q + 63) / 64, (P + 255) / 256, n_problems * p.splits), 256, 0, s>>>(p);
    else gemm_tn_kernel<<<dim3((Q + 63) / 64, (P + 63) / 64, n_problems * p.splits), 256, 0, s>>>(p);
    if (p.splits > 1) {
        const int64_t n = (int64_t)P * Q;
        gemm_tn_reduce_kernel<<<dim3((int)min((int64_t)2048, (n + 255) / 256), n_problems), 256, 0, s>>>(p);
    }
    MC_CHECK_LAUNCH();
    return 0;
}
