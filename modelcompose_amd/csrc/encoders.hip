// Encoder-specific kernels (BEATs, LanguageBind-Video, PointBERT) — all HBM/latency-bound integer/row work.
//
//   add_rows        out[r] = x[r] + table[idx[r]]      temporal embedding of LanguageBind (video/modeling_video.py:110-113),
//                                                      Q-Former audio position embedding (multimodal_projector/builder.py:136-140),
//                                                      PointBERT `x + pos` before every block (pointbert/point_encoder.py:95-98)
//   im2col_ex       strided conv operand with zero padding and a channel window (BEATs grouped pos_conv, beats/backbone.py:71-85)
//   beats_gate      GRU-style gate of the relative position bias (beats/backbone.py:689-697)
//   group_max       max over the n neighbours of each point group (pointbert/dvae.py:216-221)
//   fps             farthest point sampling, serial over npoint (pointbert/misc.py:40-60)
//   knn_group       k nearest points of every centre + centre subtraction (pointbert/dvae.py:107-141,150-187)
//   zero_rows       x[r] = 0 for masked rows (beats/backbone.py:150-151)
#include "common.h"

__global__ __launch_bounds__(256) void add_rows_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ table,
                                                       int64_t ldt, const int32_t* __restrict__ idx, bf16_t* __restrict__ out, int64_t ldo,
                                                       int D) {
    const int r = blockIdx.x;
    const int64_t ti = idx ? (int64_t)idx[r] : r;
    const bf16_t* xr = x + (int64_t)r * ldx;
    bf16_t* orow = out + (int64_t)r * ldo;
    if (ti < 0) {
        for (int c = threadIdx.x; c < (D >> 3); c += 256) *(bf16x8*)(orow + c * 8) = *(const bf16x8*)(xr + c * 8);
        return;
    }
    const bf16_t* tr = table + ti * ldt;
    for (int c = threadIdx.x; c < (D >> 3); c += 256) {
        const bf16x8 a = *(const bf16x8*)(xr + c * 8), b = *(const bf16x8*)(tr + c * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)a[j] + (float)b[j]);
        *(bf16x8*)(orow + c * 8) = o;
    }
}

extern "C" int mc_add_rows_bf16(const void* x, int64_t ldx, const void* table, int64_t ldt, const int32_t* idx, void* out, int64_t ldo,
                                int n_rows, int D, void* stream) {
    MC_CHECK_ARG(x && table && out && n_rows > 0 && D % 8 == 0 && ldx % 8 == 0 && ldt % 8 == 0 && ldo % 8 == 0, "mc_add_rows_bf16: bad arguments");
    add_rows_kernel<<<n_rows, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (const bf16_t*)table, ldt, idx, (bf16_t*)out, ldo, D);
    MC_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(256) void zero_rows_kernel(bf16_t* __restrict__ x, int64_t ldx, const int32_t* __restrict__ rows, int D) {
    bf16_t* xr = x + (int64_t)rows[blockIdx.x] * ldx;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = threadIdx.x; c < (D >> 3); c += 256) *(bf16x8*)(xr + c * 8) = z;
}

extern "C" int mc_zero_rows_bf16(void* x, int64_t ldx, const int32_t* rows, int n_rows, int D, void* stream) {
    MC_CHECK_ARG(x && (rows || n_rows == 0) && D % 8 == 0, "mc_zero_rows_bf16: bad arguments");
    if (n_rows == 0) return 0;
    zero_rows_kernel<<<n_rows, 256, 0, (hipStream_t)stream>>>((bf16_t*)x, ldx, rows, D);
    MC_CHECK_LAUNCH();
    return 0;
}

// BEATs forward_padding_mask + `x[padding_mask] = 0` (beats/BEATs.py:120-132, beats/backbone.py:150-151) WITHOUT a host round trip (round 4:
// the host-side analysis of the frame mask cost a device -> host sync at the top of every audio encode).  One workgroup per clip:
// token t is padded when ALL of its `span` frames are (mask: uint8 [B, mask_stride], the first T * span frames are used); kv_lens[b] =
// number of un-padded tokens; the rows of padded tokens are zeroed; bad[0] is set when a clip's padding is not a suffix (the attention
// kernels take one length per clip - the host raises on it at its next natural sync point).
__global__ __launch_bounds__(256) void beats_padding_kernel(const unsigned char* __restrict__ mask, int64_t mask_stride, int T, int span,
                                                            bf16_t* __restrict__ x, int64_t ldx, int D, int32_t* __restrict__ kv_lens,
                                                            int32_t* __restrict__ bad) {
    __shared__ int s_valid, s_last_valid, s_first_pad;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) { s_valid = 0; s_last_valid = -1; s_first_pad = T; }
    __syncthreads();
    const unsigned char* m = mask + (int64_t)b * mask_stride;
    for (int t = threadIdx.x; t < T; t += 256) {
        bool padded = true;
        for (int j = 0; j < span; ++j) padded = padded && (m[(int64_t)t * span + j] != 0);
        if (padded) atomicMin(&s_first_pad, t);
        else { atomicAdd(&s_valid, 1); atomicMax(&s_last_valid, t); }
    }
    __syncthreads();
    const int valid = s_valid;
    if (threadIdx.x == 0) {
        kv_lens[b] = valid;
        if (s_last_valid >= s_first_pad) atomicOr(bad, 1);       // a padded token in front of a valid one
    }
    // zero the padded rows (with trailing padding: rows valid .. T-1; in general every padded token)
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    const int chunks = D >> 3;
    for (int t = 0; t < T; ++t) {
        bool padded = true;
        for (int j = 0; j < span && padded; ++j) padded = m[(int64_t)t * span + j] != 0;
        if (!padded) continue;
        bf16_t* xr = x + ((int64_t)b * T + t) * ldx;
        for (int c = threadIdx.x; c < chunks; c += 256) *(bf16x8*)(xr + c * 8) = z;
    }
}

extern "C" int mc_beats_padding_bf16(const void* frame_mask_u8, int64_t mask_stride, int B, int T, int span, void* x, int64_t ldx, int D,
                                     int32_t* kv_lens, int32_t* bad_flag, void* stream) {
    MC_CHECK_ARG(frame_mask_u8 && x && kv_lens && bad_flag && B > 0 && T > 0 && span > 0 && D % 8 == 0 && mask_stride >= (int64_t)T * span,
                 "mc_beats_padding_bf16: bad arguments");
    beats_padding_kernel<<<B, 256, 0, (hipStream_t)stream>>>((const unsigned char*)frame_mask_u8, mask_stride, T, span, (bf16_t*)x, ldx, D, kv_lens,
                                                            bad_flag);
    MC_CHECK_LAUNCH();
    return 0;
}

// in [B, C, Hin, Win]; channels [c0, c0+Cg) ; zero padding (ph, pw); only the first oh x ow outputs are produced.
// out [B*oh*ow, Kp], column = (c*kh + i)*kw + j.
__global__ __launch_bounds__(256) void im2col_ex_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int C, int Hin,
                                                        int Win, int c0, int Cg, int kh, int kw, int sh, int sw, int ph, int pw, int oh,
                                                        int ow, int Kp, int64_t s_b, int64_t s_c, int64_t s_h, int64_t s_w) {
    const int K = Cg * kh * kw;
    const int64_t total = (int64_t)B * oh * ow * Kp;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % Kp);
        const int64_t row = i / Kp;
        bf16_t v = (bf16_t)0.0f;
        if (col < K) {
            const int j = col % kw, ii = (col / kw) % kh, c = col / (kw * kh);
            const int ox = (int)(row % ow), oy = (int)((row / ow) % oh), b = (int)(row / ((int64_t)ow * oh));
            const int y = oy * sh + ii - ph, x = ox * sw + j - pw;
            if (y >= 0 && y < Hin && x >= 0 && x < Win) v = in[b * s_b + (c0 + c) * s_c + y * s_h + x * s_w];
        }
        out[i] = v;
    }
}

extern "C" int mc_im2col_ex_bf16(const void* in, int64_t s_b, int64_t s_c, int64_t s_h, int64_t s_w, void* out, int B, int C, int Hin,
                                 int Win, int c0, int Cg, int kh, int kw, int sh, int sw, int ph, int pw, int oh, int ow, int Kp,
                                 void* stream) {
    MC_CHECK_ARG(in && out && B > 0 && Cg > 0 && c0 >= 0 && c0 + Cg <= C && oh > 0 && ow > 0 && Kp >= Cg * kh * kw, "mc_im2col_ex_bf16: bad arguments");
    const int64_t total = (int64_t)B * oh * ow * Kp;
    const int grid = (int)min((int64_t)16384, (total + 255) / 256);
    im2col_ex_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)in, (bf16_t*)out, B, C, Hin, Win, c0, Cg, kh, kw, sh, sw, ph, pw,
                                                            oh, ow, Kp, s_b, s_c, s_h, s_w);
    MC_CHECK_LAUNCH();
    return 0;
}

// g8 [(b, l, h), 8] fp32 = grep_linear(q_head); gate[b][h][l] = a*(b*grep_a[h] - 1) + 2,
// (a, b) = sigmoid(sum of the first / last 4 columns)   (beats/backbone.py:689-697)
__global__ void beats_gate_kernel(const float* __restrict__ g8, const float* __restrict__ grep_a, float* __restrict__ gate, int B, int L,
                                  int H) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * L * H) return;
    const int h = (int)(i % H), l = (int)((i / H) % L), b = (int)(i / ((int64_t)H * L));
    const float* r = g8 + i * 8;
    const float a = 1.0f / (1.0f + __expf(-(r[0] + r[1] + r[2] + r[3])));
    const float bb = 1.0f / (1.0f + __expf(-(r[4] + r[5] + r[6] + r[7])));
    gate[((int64_t)b * H + h) * L + l] = a * (bb * grep_a[h] - 1.0f) + 2.0f;
}

extern "C" int mc_beats_gate_f32(const float* g8, const float* grep_a, float* gate, int B, int L, int H, void* stream) {
    MC_CHECK_ARG(g8 && grep_a && gate && B > 0 && L > 0 && H > 0, "mc_beats_gate_f32: bad arguments");
    const int64_t n = (int64_t)B * L * H;
    beats_gate_kernel<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(g8, grep_a, gate, B, L, H);
    MC_CHECK_LAUNCH();
    return 0;
}

// out[g, 0:C] = max_i x[g*n + i, 0:C];  optionally bcast[g*n + i, 0:C] = that max (the concat of dvae.py:217-218)
__global__ __launch_bounds__(256) void group_max_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ out, int64_t ldo,
                                                        bf16_t* __restrict__ bcast, int64_t ldb, int n, int C) {
    const int g = blockIdx.x;
    for (int c = threadIdx.x; c < (C >> 3); c += 256) {
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
        for (int i = 0; i < n; ++i) {
            const bf16x8 v = *(const bf16x8*)(x + ((int64_t)g * n + i) * ldx + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], (float)v[j]);
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)m[j];
        if (out) *(bf16x8*)(out + (int64_t)g * ldo + c * 8) = o;
        if (bcast)
            for (int i = 0; i < n; ++i) *(bf16x8*)(bcast + ((int64_t)g * n + i) * ldb + c * 8) = o;
    }
}

extern "C" int mc_group_max_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, void* bcast, int64_t ldb, int G, int n, int C,
                                 void* stream) {
    MC_CHECK_ARG(x && (out || bcast) && G > 0 && n > 0 && C % 8 == 0, "mc_group_max_bf16: bad arguments");
    group_max_kernel<<<G, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, ldx, (bf16_t*)out, ldo, (bf16_t*)bcast, ldb, n, C);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Farthest point sampling: one workgroup (1024 threads) per cloud, running min distances in registers.
// pts [B, N, C] bf16 (xyz first); fp32 arithmetic without contraction so the argmax sequence matches a
// fp32 torch evaluation of  sum((xyz - centroid)**2, -1)  exactly; ties -> lowest index.
#define FPS_T 1024
#define FPS_MAXP 16
__global__ __launch_bounds__(FPS_T) void fps_kernel(const bf16_t* __restrict__ pts, int N, int C, const int32_t* __restrict__ start,
                                                    int npoint, int32_t* __restrict__ out_idx, float* __restrict__ centers) {
    __shared__ float sv[FPS_T / 64];
    __shared__ int si[FPS_T / 64];
    __shared__ int s_far;
    const int b = blockIdx.x, tid = threadIdx.x;
    const bf16_t* P = pts + (int64_t)b * N * C;
    float px[FPS_MAXP], py[FPS_MAXP], pz[FPS_MAXP], dmin[FPS_MAXP];
    const int per = (N + FPS_T - 1) / FPS_T;
#pragma unroll
    for (int i = 0; i < FPS_MAXP; ++i) {
        const int n = i * FPS_T + tid;
        if (i < per && n < N) {
            px[i] = (float)P[(int64_t)n * C]; py[i] = (float)P[(int64_t)n * C + 1]; pz[i] = (float)P[(int64_t)n * C + 2];
        } else { px[i] = py[i] = pz[i] = 0.f; }
        dmin[i] = 1e10f;
    }
    int far = start ? start[b] : 0;
    for (int it = 0; it < npoint; ++it) {
        const float cx = (float)P[(int64_t)far * C], cy = (float)P[(int64_t)far * C + 1], cz = (float)P[(int64_t)far * C + 2];
        if (tid == 0) {
            out_idx[(int64_t)b * npoint + it] = far;
            if (centers) { float* c = centers + ((int64_t)b * npoint + it) * 3; c[0] = cx; c[1] = cy; c[2] = cz; }
        }
        float best = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < FPS_MAXP; ++i) {
            const int n = i * FPS_T + tid;
            if (i < per && n < N) {
                const float dx = __fsub_rn(px[i], cx), dy = __fsub_rn(py[i], cy), dz = __fsub_rn(pz[i], cz);
                const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                dmin[i] = fminf(dmin[i], d);
                if (dmin[i] > best || (dmin[i] == best && n < bi)) { best = dmin[i]; bi = n; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(best, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
        }
        if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < FPS_T / 64; ++w)
                if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
            s_far = bi;
        }
        __syncthreads();
        far = s_far;
    }
}

extern "C" int mc_fps_bf16(const void* pts, int B, int N, int C, const int32_t* start_idx, int npoint, int32_t* out_idx, float* centers,
                           void* stream) {
    MC_CHECK_ARG(pts && out_idx && B > 0 && N > 0 && C >= 3 && npoint > 0, "mc_fps_bf16: bad arguments");
    MC_CHECK_ARG(N <= FPS_T * FPS_MAXP, "mc_fps_bf16: at most %d points per cloud (got %d)", FPS_T * FPS_MAXP, N);
    fps_kernel<<<B, FPS_T, 0, (hipStream_t)stream>>>((const bf16_t*)pts, N, C, start_idx, npoint, out_idx, centers);
    MC_CHECK_LAUNCH();
    return 0;
}

// k nearest neighbours of each centre (squared distance, fp32), ascending; writes the neighbourhood rows
// [ (b, g, j), Kp ] bf16 = [xyz - centre | other features | 0-pad] ready to be a GEMM operand, and the indices.
#define KNN_T 256
__global__ __launch_bounds__(KNN_T) void knn_group_kernel(const bf16_t* __restrict__ pts, int N, int C, const float* __restrict__ centers,
                                                          int G, int k, bf16_t* __restrict__ out, int Kp, int32_t* __restrict__ out_idx) {
    extern __shared__ float dist[];          // N floats
    __shared__ float sv[KNN_T / 64];
    __shared__ int si[KNN_T / 64];
    __shared__ int s_sel;
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const bf16_t* P = pts + (int64_t)b * N * C;
    const float* c = centers + ((int64_t)b * G + g) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    for (int n = tid; n < N; n += KNN_T) {
        const float dx = __fsub_rn((float)P[(int64_t)n * C], cx), dy = __fsub_rn((float)P[(int64_t)n * C + 1], cy),
                    dz = __fsub_rn((float)P[(int64_t)n * C + 2], cz);
        dist[n] = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        float best = INFINITY;
        int bi = 0x7fffffff;
        for (int n = tid; n < N; n += KNN_T) {
            const float d = dist[n];
            if (d < best || (d == best && n < bi)) { best = d; bi = n; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(best, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (v2 < best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
        }
        if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < KNN_T / 64; ++w)
                if (sv[w] < best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
            s_sel = bi;
            dist[bi] = INFINITY;
            if (out_idx) out_idx[((int64_t)b * G + g) * k + j] = bi;
        }
        __syncthreads();
        const int sel = s_sel;
        bf16_t* orow = out + (((int64_t)b * G + g) * k + j) * Kp;
        for (int col = tid; col < Kp; col += KNN_T) {
            float v = 0.f;
            if (col < C) {
                v = (float)P[(int64_t)sel * C + col];
                if (col == 0) v -= cx; else if (col == 1) v -= cy; else if (col == 2) v -= cz;
            }
            orow[col] = (bf16_t)v;
        }
    }
}

extern "C" int mc_knn_group_bf16(const void* pts, int B, int N, int C, const float* centers, int G, int k, void* out, int Kp,
                                 int32_t* out_idx, void* stream) {
    MC_CHECK_ARG(pts && centers && out && B > 0 && N >= k && C >= 3 && G > 0 && k > 0 && Kp >= C, "mc_knn_group_bf16: bad arguments");
    MC_CHECK_ARG((size_t)N * 4 <= 150 * 1024, "mc_knn_group_bf16: at most 38400 points per cloud (got %d)", N);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)knn_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; }
    knn_group_kernel<<<dim3(G, B), KNN_T, (size_t)N * 4, (hipStream_t)stream>>>((const bf16_t*)pts, N, C, centers, G, k, (bf16_t*)out, Kp,
                                                                               out_idx);
    MC_CHECK_LAUNCH();
    return 0;
}

// centres (fp32 [rows, 3]) -> bf16 rows padded to Kp (GEMM operand of pos_embed, pointbert/point_encoder.py:140-144)
__global__ void f32_rows_to_bf16_kernel(const float* __restrict__ in, int C, bf16_t* __restrict__ out, int Kp, int64_t rows) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= rows * Kp) return;
    const int col = (int)(i % Kp);
    out[i] = col < C ? (bf16_t)in[(i / Kp) * C + col] : (bf16_t)0.0f;
}

extern "C" int mc_f32_rows_to_bf16(const float* in, int C, void* out, int Kp, int64_t rows, void* stream) {
    MC_CHECK_ARG(in && out && rows > 0 && Kp >= C, "mc_f32_rows_to_bf16: bad arguments");
    f32_rows_to_bf16_kernel<<<(int)((rows * Kp + 255) / 256), 256, 0, (hipStream_t)stream>>>(in, C, (bf16_t*)out, Kp, rows);
    MC_CHECK_LAUNCH();
    return 0;
}
