// Softmax-attention backward for gfx950 (training step, BASELINE config 5).
//
// Replaces autograd through LocalLoraAttention.forward's  softmax_fp32(QK^T/sqrt(d) + mask)·V
// (modelcompose/model/language_model/multimodal_llama.py:295-312) and the flash-attn backward the reference uses in training
// (modelcompose/train/multimodal_flash_attn_monkey_patch.py:96-106, third-party CUDA kernel).  Flash-style: P is recomputed
// from Q, K and the forward's log-sum-exp, no LxS matrix in HBM.  With s = scale·q·k, p = softmax(s), dp = dO·V^T,
// delta = rowsum(dO ⊙ O):   ds = p ⊙ (dp − delta),  dQ = scale·ds·K,  dK = scale·ds^T·Q,  dV = p^T·dO.
//
//   attn_delta_kernel : delta[b,h,t] = sum_d dO·O
//   attn_bwd_dq_kernel  : one workgroup = 4 waves x 16 queries of one (batch, head), loops over key tiles of 64.
//       Same operand orientation as the forward kernel: S^T = K·Q^T and dP^T = V·dO^T put one query on a lane, so
//       lse / delta are lane scalars and dS^T feeds  dQ^T += K^T·dS^T  straight from registers.
//   attn_bwd_dkv_kernel : one workgroup = 4 waves x 16 keys, loops over query tiles.  S = Q·K^T and dP = dO·V^T put one
//       key on a lane; P and dS feed  dV^T += dO^T·P  and  dK^T += Q^T·dS  from registers.
// One XOR-swizzled LDS copy per tile serves both the row reads (ds_read_b128) and the transposed reads (ds_read_b64_tr_b16, tr_off).
#include "common.h"

#define NEG_BIG (-1.0e30f)

// v_exp_f32 directly: the arguments are (score - running max) <= 0 or (old max - new max) <= 0, so the range fix-up that exp2f()
// expands to (compare, scale, ldexp: ~4 extra VALU instructions per call, 17 calls per K/V tile) is dead weight
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttnBwdParams {
    const bf16_t* q; int64_t q_sb, q_st, q_sh;      // [B, Lq, H, D]-like strides (elements)
    const bf16_t* k; int64_t k_sb, k_st, k_sh;
    const bf16_t* v; int64_t v_sb, v_st, v_sh;
    const bf16_t* o; const bf16_t* d_o; int64_t o_sb, o_st, o_sh;     // O and dO share strides
    const float* lse;                               // [B, H, Lq] log2 domain (mc_attn_prefill_lse_bf16)
    float* delta;                                   // [B, H, Lq]
    bf16_t* dq; int64_t dq_sb, dq_st, dq_sh;
    bf16_t* dk; int64_t dk_sb, dk_st, dk_sh;
    bf16_t* dv; int64_t dv_sb, dv_st, dv_sh;
    const int32_t* kv_lens;
    int B, H, Lq, S, causal, q_offset;
    float scale, scale_log2e;
    AttnDropout drop;                               // DROP instantiations: the forward's dropout on the probabilities, regenerated
};

template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnBwdParams p) {
    // D/8 lanes per (b, h, t) row, 64 / (D/8) rows per wave: every lane loads 16 bytes of O and of dO
    constexpr int LPR = D / 8, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
    const int total = p.B * p.H * p.Lq;
    const bool ok = row < total;
    const int rr = ok ? row : 0;
    const int t = rr % p.Lq, h = (rr / p.Lq) % p.H, b = rr / (p.Lq * p.H);
    const int64_t off = b * p.o_sb + t * p.o_st + h * p.o_sh + (lane % LPR) * 8;
    const bf16x8 a = *(const bf16x8*)(p.o + off), d = *(const bf16x8*)(p.d_o + off);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)d[j];
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);
    if (ok && lane % LPR == 0) p.delta[row] = s;
}

template <int D>
__device__ __forceinline__ int swz(int ch, int row) {
    return (D == 128) ? (ch ^ (row & 15)) : (ch ^ ((row >> 1) & 7));
}

// byte offset, inside a swizzled tile, of the 8 bytes a transposing read (ds_read_b64_tr_b16) needs for logical column block `db`
// (16 columns) / quarter `tp` (4 columns) of `row`: the 16-byte chunk is XOR-swizzled exactly as the row reads expect, so one LDS copy
// of a tile serves both access patterns (and the 8 rows of one read group land on distinct banks, as in the forward kernel's V reads)
template <int D>
__device__ __forceinline__ int tr_off(int row, int db, int tp) {
    return row * (D * 2) + swz<D>(db * 2 + (tp >> 1), row) * 16 + (tp & 1) * 8;
}

// stage 64 rows x D of src (row stride in elements) into a swizzled and/or linear LDS tile; rows >= n_valid are clamped
template <int D, bool SW, bool LIN>
__device__ __forceinline__ void stage_tile(const bf16_t* src, int64_t row_stride, int row0, int n_rows_total, char* sw_tile, char* lin_tile,
                                           int tid) {
    constexpr int ROWB = D * 2, CH = ROWB / 16;
#pragma unroll
    for (int it = 0; it < (64 * CH) / 256; ++it) {
        const int idx = it * 256 + tid;
        const int row = idx / CH, ch = idx % CH;
        const int r = min(row0 + row, n_rows_total - 1);
        const u32x4 v4 = *(const u32x4*)(src + (int64_t)r * row_stride + ch * 8);
        if (SW) *(u32x4*)(sw_tile + row * ROWB + swz<D>(ch, row) * 16) = v4;
        if (LIN) *(u32x4*)(lin_tile + row * ROWB + ch * 16) = v4;
    }
}

// DROP (forward ran mc_attn_prefill_dropout_bf16): O = (P ⊙ m) V with m = keep / (1 - p), so dP reaches the softmax through the same
// mask: ds = p ⊙ (m ⊙ dp − delta) (delta = rowsum(dO ⊙ O) of the dropped output, unchanged), dV = (p ⊙ m)^T dO
template <int D, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnBwdParams p) {
    constexpr int ROWB = D * 2, KS = D / 32, DB = D / 16;
    __shared__ __attribute__((aligned(16))) char lds[2 * 64 * ROWB];
    char* k_sw = lds;
    char* v_sw = lds + 64 * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 64 + wave * 16;
    const int kvlen = p.kv_lens ? min(p.kv_lens[b], p.S) : p.S;
    const int t = min(q0 + c, p.Lq - 1);

    bf16x8 qf[KS], dof[KS];
    {
        const bf16_t* qp = p.q + b * p.q_sb + t * p.q_st + h * p.q_sh + g * 8;
        const bf16_t* dp_ = p.d_o + b * p.o_sb + t * p.o_st + h * p.o_sh + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { qf[ks] = *(const bf16x8*)(qp + ks * 32); dof[ks] = *(const bf16x8*)(dp_ + ks * 32); }
    }
    const int64_t stat = ((int64_t)b * p.H + h) * p.Lq + t;
    const float lse = p.lse[stat], delta = p.delta[stat];
    const int q_abs = q0 + c + p.q_offset;

    f32x4 acc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int last_key = kvlen;
    if (p.causal) last_key = min(last_key, blockIdx.x * 64 + 63 + p.q_offset + 1);
    const int ntiles = (last_key + 63) / 64;
    const bf16_t* kbase = p.k + b * p.k_sb + h * p.k_sh;
    const bf16_t* vbase = p.v + b * p.v_sb + h * p.v_sh;

    // the next key tile (K, V) travels from global memory into registers under the MFMAs of the current one
    constexpr int CHB = ROWB / 16, NLD = (64 * CHB) / 256;
    u32x4 pk[NLD], pv[NLD];
    auto gload = [&](int kt) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / CHB, ch = idx % CHB;
            const int r = min(kt * 64 + row, p.S - 1);
            pk[it] = *(const u32x4*)(kbase + (int64_t)r * p.k_st + ch * 8);
            pv[it] = *(const u32x4*)(vbase + (int64_t)r * p.v_st + ch * 8);
        }
    };
    if (ntiles > 0) gload(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / CHB, ch = idx % CHB;
            *(u32x4*)(k_sw + row * ROWB + swz<D>(ch, row) * 16) = pk[it];
            *(u32x4*)(v_sw + row * ROWB + swz<D>(ch, row) * 16) = pv[it];
        }
        __syncthreads();
        if (kt + 1 < ntiles) gload(kt + 1);
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            s[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = kb * 16 + c;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = row * ROWB + swz<D>(ks * 4 + g, row) * 16;
                const bf16x8 kf = *(const bf16x8*)(k_sw + off);
                const bf16x8 vf = *(const bf16x8*)(v_sw + off);
                s[kb] = mc_mfma_16x16x32(kf, qf[ks], s[kb]);
                dp[kb] = mc_mfma_16x16x32(vf, dof[ks], dp[kb]);
            }
        }
        bf16x8 dsf[2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if constexpr (DROP) attn_dropout_quad(p.drop, stat, p.S, kt * 64 + kb * 16 + g * 4, dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 64 + kb * 16 + g * 4 + r;
                const bool ok = key < kvlen && (!p.causal || key <= q_abs);
                const float pv = ok ? fast_exp2(s[kb][r] * p.scale_log2e - lse) : 0.f;
                dsf[kb >> 1][(kb & 1) * 4 + r] = (bf16_t)(pv * ((DROP ? dp[kb][r] * dm[r] : dp[kb][r]) - delta));
            }
        }
        // dQ^T[d][query] += K^T · dS^T   (K^T through the transposing LDS read of the linear tile)
        const int tq = (lane & 15) >> 2, tp = lane & 3;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int key_lo = (2 * pr) * 16 + g * 4 + tq;
            const int key_hi = (2 * pr + 1) * 16 + g * 4 + tq;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const bf16x4 lo = mc_ds_read_tr16(
                    (__attribute__((address_space(3))) bf16x4*)(k_sw + tr_off<D>(key_lo, db, tp)));
                const bf16x4 hi = mc_ds_read_tr16(
                    (__attribute__((address_space(3))) bf16x4*)(k_sw + tr_off<D>(key_hi, db, tp)));
                bf16x8 kf;
                kf[0] = lo[0]; kf[1] = lo[1]; kf[2] = lo[2]; kf[3] = lo[3];
                kf[4] = hi[0]; kf[5] = hi[1]; kf[6] = hi[2]; kf[7] = hi[3];
                acc[db] = mc_mfma_16x16x32(kf, dsf[pr], acc[db]);
            }
        }
    }
    if (q0 + c < p.Lq) {
        bf16_t* op = p.dq + b * p.dq_sb + (int64_t)(q0 + c) * p.dq_st + h * p.dq_sh;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            bf16x4 ov = {(bf16_t)(acc[db][0] * p.scale), (bf16_t)(acc[db][1] * p.scale), (bf16_t)(acc[db][2] * p.scale),
                         (bf16_t)(acc[db][3] * p.scale)};
            *(bf16x4*)(op + db * 16 + g * 4) = ov;
        }
    }
}

template <int D, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnBwdParams p) {
    constexpr int ROWB = D * 2, KS = D / 32, DB = D / 16;
    __shared__ __attribute__((aligned(16))) char lds[2 * 64 * ROWB + 2 * 64 * 4];
    char* q_sw = lds;
    char* o_sw = lds + 64 * ROWB;
    float* lse_t = (float*)(lds + 2 * 64 * ROWB);
    float* del_t = lse_t + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int k0 = blockIdx.x * 64 + wave * 16;
    const int kvlen = p.kv_lens ? min(p.kv_lens[b], p.S) : p.S;
    const int key = k0 + c;                       // this lane's key column
    const int keyc = min(key, p.S - 1);

    bf16x8 kf[KS], vf[KS];
    {
        const bf16_t* kp = p.k + b * p.k_sb + (int64_t)keyc * p.k_st + h * p.k_sh + g * 8;
        const bf16_t* vp = p.v + b * p.v_sb + (int64_t)keyc * p.v_st + h * p.v_sh + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { kf[ks] = *(const bf16x8*)(kp + ks * 32); vf[ks] = *(const bf16x8*)(vp + ks * 32); }
    }
    f32x4 acck[DB], accv[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i) { acck[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; accv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    // causal: query t (absolute t + q_offset) sees key iff key <= t + q_offset  ->  first query tile that can see this key tile
    int first_q = 0;
    if (p.causal) first_q = max(0, blockIdx.x * 64 - p.q_offset) / 64;
    const int nq = (p.Lq + 63) / 64;
    const bf16_t* qbase = p.q + b * p.q_sb + h * p.q_sh;
    const bf16_t* dobase = p.d_o + b * p.o_sb + h * p.o_sh;
    const int64_t stat0 = ((int64_t)b * p.H + h) * p.Lq;

    // the next query tile (Q, dO, lse, delta) travels from global memory into registers under the MFMAs of the current one
    constexpr int CHB = ROWB / 16, NLD = (64 * CHB) / 256;
    u32x4 pq[NLD], po[NLD];
    float plse = 0.f, pdel = 0.f;
    auto gload = [&](int qt) {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / CHB, ch = idx % CHB;
            const int r = min(qt * 64 + row, p.Lq - 1);
            pq[it] = *(const u32x4*)(qbase + (int64_t)r * p.q_st + ch * 8);
            po[it] = *(const u32x4*)(dobase + (int64_t)r * p.o_st + ch * 8);
        }
        if (tid < 64) {
            const int tt = min(qt * 64 + tid, p.Lq - 1);
            plse = p.lse[stat0 + tt];
            pdel = p.delta[stat0 + tt];
        }
    };
    if (first_q < nq) gload(first_q);
    for (int qt = first_q; qt < nq; ++qt) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = it * 256 + tid;
            const int row = idx / CHB, ch = idx % CHB;
            *(u32x4*)(q_sw + row * ROWB + swz<D>(ch, row) * 16) = pq[it];
            *(u32x4*)(o_sw + row * ROWB + swz<D>(ch, row) * 16) = po[it];
        }
        if (tid < 64) { lse_t[tid] = plse; del_t[tid] = pdel; }
        __syncthreads();
        if (qt + 1 < nq) gload(qt + 1);
        f32x4 s[4], dp[4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            s[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dp[qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = qb * 16 + c;             // A operand row = query
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = row * ROWB + swz<D>(ks * 4 + g, row) * 16;
                const bf16x8 qf = *(const bf16x8*)(q_sw + off);
                const bf16x8 of = *(const bf16x8*)(o_sw + off);
                s[qb] = mc_mfma_16x16x32(qf, kf[ks], s[qb]);
                dp[qb] = mc_mfma_16x16x32(of, vf[ks], dp[qb]);
            }
        }
        // lane owns key column c, queries 16qb + 4g + r
        bf16x8 pf[2], dsf[2];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = qb * 16 + g * 4 + r;
                const int tq_ = qt * 64 + ql;
                const bool ok = tq_ < p.Lq && key < kvlen && (!p.causal || key <= tq_ + p.q_offset);
                const float pv = ok ? fast_exp2(s[qb][r] * p.scale_log2e - lse_t[ql]) : 0.f;
                // this lane owns one key and four consecutive QUERIES: one mask quad per element
                const float dm = DROP ? attn_dropout_one(p.drop, stat0 + min(tq_, p.Lq - 1), p.S, keyc) : 1.f;
                pf[qb >> 1][(qb & 1) * 4 + r] = (bf16_t)(DROP ? pv * dm : pv);
                dsf[qb >> 1][(qb & 1) * 4 + r] = (bf16_t)(pv * ((DROP ? dp[qb][r] * dm : dp[qb][r]) - del_t[ql]));
            }
        const int tq = (lane & 15) >> 2, tp = lane & 3;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int r_lo = (2 * pr) * 16 + g * 4 + tq;
            const int r_hi = (2 * pr + 1) * 16 + g * 4 + tq;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const int off_lo = tr_off<D>(r_lo, db, tp), off_hi = tr_off<D>(r_hi, db, tp);
                const bf16x4 olo = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(o_sw + off_lo));
                const bf16x4 ohi = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(o_sw + off_hi));
                const bf16x4 qlo = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(q_sw + off_lo));
                const bf16x4 qhi = mc_ds_read_tr16((__attribute__((address_space(3))) bf16x4*)(q_sw + off_hi));
                bf16x8 of, qf;
                of[0] = olo[0]; of[1] = olo[1]; of[2] = olo[2]; of[3] = olo[3]; of[4] = ohi[0]; of[5] = ohi[1]; of[6] = ohi[2]; of[7] = ohi[3];
                qf[0] = qlo[0]; qf[1] = qlo[1]; qf[2] = qlo[2]; qf[3] = qlo[3]; qf[4] = qhi[0]; qf[5] = qhi[1]; qf[6] = qhi[2]; qf[7] = qhi[3];
                accv[db] = mc_mfma_16x16x32(of, pf[pr], accv[db]);
                acck[db] = mc_mfma_16x16x32(qf, dsf[pr], acck[db]);
            }
        }
    }
    if (key < p.S) {
        bf16_t* kp = p.dk + b * p.dk_sb + (int64_t)key * p.dk_st + h * p.dk_sh;
        bf16_t* vp = p.dv + b * p.dv_sb + (int64_t)key * p.dv_st + h * p.dv_sh;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            bf16x4 ok_ = {(bf16_t)(acck[db][0] * p.scale), (bf16_t)(acck[db][1] * p.scale), (bf16_t)(acck[db][2] * p.scale),
                          (bf16_t)(acck[db][3] * p.scale)};
            bf16x4 ov = {(bf16_t)accv[db][0], (bf16_t)accv[db][1], (bf16_t)accv[db][2], (bf16_t)accv[db][3]};
            *(bf16x4*)(kp + db * 16 + g * 4) = ok_;
            *(bf16x4*)(vp + db * 16 + g * 4) = ov;
        }
    }
}

extern "C" int mc_attn_bwd_bf16(const mc_attn_bwd_args* a, void* stream) {
    MC_CHECK_ARG(a && a->q && a->k && a->v && a->o && a->d_o && a->lse && a->delta && a->dq && a->dk && a->dv, "mc_attn_bwd_bf16: null pointer");
    MC_CHECK_ARG(a->D == 64 || a->D == 128, "mc_attn_bwd_bf16: head_dim %d not supported (64 or 128)", a->D);
    MC_CHECK_ARG(a->B > 0 && a->H > 0 && a->Lq > 0 && a->S > 0, "mc_attn_bwd_bf16: bad shape");
    AttnBwdParams p{(const bf16_t*)a->q, a->q_sb, a->q_st, a->q_sh, (const bf16_t*)a->k, a->k_sb, a->k_st, a->k_sh,
                    (const bf16_t*)a->v, a->v_sb, a->v_st, a->v_sh, (const bf16_t*)a->o, (const bf16_t*)a->d_o, a->o_sb, a->o_st, a->o_sh,
                    a->lse, a->delta, (bf16_t*)a->dq, a->dq_sb, a->dq_st, a->dq_sh, (bf16_t*)a->dk, a->dk_sb, a->dk_st, a->dk_sh,
                    (bf16_t*)a->dv, a->dv_sb, a->dv_st, a->dv_sh, a->kv_lens, a->B, a->H, a->Lq, a->S, a->causal, a->q_offset,
                    a->scale, a->scale * 1.4426950408889634f};
    const bool drop = a->dropout_p > 0.f;
    if (drop) {
        MC_CHECK_ARG(a->dropout_p < 1.f && a->S % 4 == 0, "mc_attn_bwd_bf16: dropout needs 0 <= p < 1 and S %% 4 == 0");
        const double t = (double)a->dropout_p * 4294967296.0;
        p.drop = AttnDropout{t >= 4294967295.0 ? 4294967295u : (uint32_t)t, (uint32_t)a->dropout_seed, (uint32_t)(a->dropout_seed >> 32),
                             a->dropout_stream, 1.0f / (1.0f - a->dropout_p)};
    }
    hipStream_t s = (hipStream_t)stream;
    const int rows = a->B * a->H * a->Lq;
    dim3 gq((a->Lq + 63) / 64, a->H, a->B), gk((a->S + 63) / 64, a->H, a->B);
    if (a->D == 128) {
        attn_delta_kernel<128><<<(rows + 15) / 16, 256, 0, s>>>(p);                 // 4 waves x 4 rows
        if (drop) { attn_bwd_dq_kernel<128, true><<<gq, 256, 0, s>>>(p); attn_bwd_dkv_kernel<128, true><<<gk, 256, 0, s>>>(p); }
        else { attn_bwd_dq_kernel<128, false><<<gq, 256, 0, s>>>(p); attn_bwd_dkv_kernel<128, false><<<gk, 256, 0, s>>>(p); }
    } else {
        attn_delta_kernel<64><<<(rows + 31) / 32, 256, 0, s>>>(p);                  // 4 waves x 8 rows
        if (drop) { attn_bwd_dq_kernel<64, true><<<gq, 256, 0, s>>>(p); attn_bwd_dkv_kernel<64, true><<<gk, 256, 0, s>>>(p); }
        else { attn_bwd_dq_kernel<64, false><<<gq, 256, 0, s>>>(p); attn_bwd_dkv_kernel<64, false><<<gk, 256, 0, s>>>(p); }
    }
    MC_CHECK_LAUNCH();
    return 0;
}
