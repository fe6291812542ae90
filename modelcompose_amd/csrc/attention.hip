// Softmax attention for gfx950 (bf16 storage, fp32 softmax / accumulation).
//
// Replaces LocalLoraAttention.forward's  softmax_fp32(QK^T/sqrt(d) + mask)·V
// (modelcompose/model/language_model/multimodal_llama.py:295-312; the additive causal+padding
// mask of transformers==4.31 LlamaModel._prepare_decoder_attention_mask becomes an index test)
// and the bidirectional CLIPAttention of the vision/video towers.
//
//   attn_prefill_kernel : flash-style, no LxS score matrix in HBM.  One workgroup = 4 waves = 64 query
//       rows of one (batch, head); K/V tiles of 64 keys are staged in LDS (K XOR-swizzled for
//       conflict-free ds_read_b128, V row-major and consumed through ds_read_b64_tr_b16).
//       Scores are computed transposed (S^T = K·Q^T) so each lane owns one query column: the row
//       statistics are lane-local and P feeds the second MFMA (O^T = V^T·P^T) straight from registers.
//   attn_decode_kernel  : one query row per (batch, head), HBM-bound KV stream straight to VGPRs,
//       split over the KV length (flash-decoding) + a small combine kernel.
#include "common.h"
#include <stdlib.h>

struct AttnParams {
    const bf16_t* q; int64_t q_sb, q_st, q_sh;     // strides in elements; head_dim contiguous
    const bf16_t* k; int64_t k_sb, k_st, k_sh;
    const bf16_t* v; int64_t v_sb, v_st, v_sh;
    bf16_t* o; int64_t o_row_stride;               // out[row*o_row_stride + h*D + d]
    const int32_t* out_map;                        // optional [B*Lq] -> output row (-1 = skip)
    const int32_t* kv_lens;                        // optional [B] valid keys per batch entry
    int B, H, Hkv, Lq, S;
    int causal, q_offset;                          // query t has absolute position t + q_offset
    float scale_log2e;                             // softmax scale * log2(e)
    // optional gated relative-position bias (BEATs, beats/backbone.py:431-468,689-697):
    //   score += q_gate[b,h,t] * rel_table[h*rel_stride + (key - t + rel_off)]
    const float* rel_table; const float* q_gate; int rel_stride, rel_off;
    float* lse;                                    // optional [B, H, Lq]: log2-sum-exp of the scaled scores (training: backward input)
    AttnDropout drop;                              // DROP instantiations only: dropout on the probabilities (training, Q-Former projector)
    // optional per-key validity [B][key_valid_sb] (0 = the key is masked for every query): attention masks with zeros that are not a
    // suffix - left padding, holes (the additive padding mask of LlamaModel._prepare_decoder_attention_mask, multimodal_llama.py:543-545)
    const uint8_t* key_valid; int64_t key_valid_sb;
    // optional two-level batch index (attn_tiny_kernel only; mc_attn_set_batch_split): batch entry b = (b / b_inner, b % b_inner) sits at
    // (b / b_inner) * *_sb + (b % b_inner) * sbi - sequences that interleave in memory (the temporal attention of LanguageBind-Video over
    // the (b t) n d layout: sequence (b, n), tokens a frame apart) are attended in place, without a permuted copy
    int b_inner; int64_t sbi;
};

#define NEG_BIG (-1.0e30f)

// v_exp_f32 directly: the arguments are (score - running max) <= 0 or (old max - new max) <= 0, so the range fix-up that exp2f()
// expands to (compare, scale, ldexp: ~4 extra VALU instructions per call, 17 calls per K/V tile) is dead weight
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max / sum over the four 16-lane rows of a wave (lanes c, c+16, c+32, c+48 share a query column).  v_permlane16_swap / v_permlane32_swap
// are VALU instructions; __shfl_xor(x, 16 / 32) compiles to ds_bpermute_b32, an LDS round trip (~150 cycles each) on the critical path of
// the online softmax: four dependent ones per tile and query block.
__device__ __forceinline__ float rows4_max(float t) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    t = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows4_sum(float t) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    t = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// acc *= alpha IN PLACE (asm with tied operands): the rare rescale branch of the online softmax must leave the accumulators in the
// registers they live in (see the rescale below)
__device__ __forceinline__ void scale_inplace(f32x4& a, float alpha) {
    asm volatile("v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(alpha));
}

// One LDS-DMA instruction (64 lanes x 16 bytes -> 1 KiB at the wave-uniform LDS address).  Issued from inline asm on purpose: hipcc
// otherwise orders every later ds_read behind it with s_waitcnt vmcnt(0) (it cannot prove the read does not alias the DMA
// destination), which serialises the prefetch of the next K/V tile with the MFMAs of the current one.  The kernel waits for these
// transfers itself (s_waitcnt vmcnt(0) + s_barrier before the first read of a buffer).
__device__ __forceinline__ void dma16(const void* gptr, const void* lds_dst) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(m0v) : "memory");
}

// Same, scalar-base form: address = wave-uniform 64-bit base (SGPRs) + per-lane 32-bit byte offset.  The offsets of a K/V tile are the
// same for every tile (row-in-tile * stride + swizzled chunk), so the loop needs no vector address arithmetic at all: only the base moves.
__device__ __forceinline__ void dma16s(const void* base_uniform, uint32_t off, const void* lds_dst) {
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds_dst);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base_uniform), "s"(m0v) : "memory");
}
// ... with the LDS destination as an integer address computed on the scalar unit (a generic pointer costs a null check and a 64-bit
// aperture round trip per instruction: 6 scalar instructions, 8 times per tile and wave)
__device__ __forceinline__ void dma16si(const void* base_uniform, uint32_t off, uint32_t lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base_uniform), "s"(lds_addr_uniform) : "memory");
}

// REL: gated relative-position bias of BEATs compiled in (a separate instantiation keeps its per-score loads and branches out of the
// common kernel)
// QW: waves per workgroup (4 or 8).  8 waves share each staged K/V tile between twice the queries, which halves the LDS fills, DMA issues
// and barriers per query and puts 4 waves on every SIMD at two workgroups per CU.
// NQ: 16-query blocks per wave (1 or 2).  A wave reads the WHOLE staged K and V tile from LDS (16 KiB each at D = 128) for every tile; with
// one query block that is 32 KiB of LDS reads behind 32 MFMAs (512 cycles): four computing waves per CU ask for 256 B/clk, the LDS's peak
// - the kernel is LDS-bandwidth bound at ~25 % MFMA utilisation.  With NQ = 2 every K / V fragment read feeds two MFMAs (one per query
// block): half the LDS bytes per MFMA.
// DROP: dropout on the attention probabilities (BertSelfAttention.dropout, Qformer.py:259): P keeps its fp32 row sum (softmax is
// normalised BEFORE the dropout in the reference) and enters P.V as bf16(P * keep / (1 - p)); masks from Philox (common.h)
template <int D, bool REL, int QW, int NQ, bool DROP = false, int LATE = 0, bool KM = true, bool CL = true>
__global__ __launch_bounds__(64 * QW, 2) void attn_prefill_kernel(AttnParams p) {
    constexpr int ROWB = D * 2;              // bytes per K/V row
    constexpr int CH = ROWB / 16;            // 16-byte chunks per row
    constexpr int KS = D / 32;               // MFMA k-steps over the head dim
    constexpr int DB = D / 16;               // 16-wide d blocks of the output
    constexpr int TILE = 64 * ROWB;          // bytes of one staged K (or V) tile
    constexpr int RPI = 1024 / ROWB;         // rows written by one 1-KiB LDS-DMA wave instruction
    constexpr int WROWS = 64 / QW;           // rows of a K/V tile staged by one wave
    constexpr int NDMA = WROWS / RPI;        // DMA instructions per wave per operand per tile
    constexpr int WQ = 16 * NQ;              // query rows per wave
    constexpr int QB = WQ * QW;              // query rows per workgroup
    extern __shared__ __attribute__((aligned(16))) char lds[];               // 2 * 2 * 64 * ROWB bytes: [buffer][K | V]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    // XCD-aware block order.  Workgroups are dealt to the 8 XCDs round-robin by linear id, so the query blocks of one (batch, head) - which
    // all stream the same K/V - would land on 8 different L2s and each would fetch that K/V from the fabric again.  Give every XCD a
    // contiguous range of the logical order instead (query block fastest): one (batch, head) stays on one XCD and its K/V tiles are L2
    // hits after the first reader.  Within a (batch, head) the causal query blocks run longest-first.
    int qblk, h, b;
    {
        const int nx = gridDim.x, ny = gridDim.y;
        const int total = nx * ny * (int)gridDim.z;
        const int lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
        const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7, idx = lin >> 3;
        const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        qblk = nx - 1 - v % nx;
        h = (v / nx) % ny;
        b = v / (nx * ny);
    }
    const int hk = h / (p.H / p.Hkv);
    const int q0 = qblk * QB + wave * WQ;
    const int kvlen = p.kv_lens ? min(p.kv_lens[b], p.S) : p.S;

    // Q fragments (B operand: col = query c, k = d)
    bf16x8 qf[NQ][KS];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        const int t = min(q0 + nq * 16 + c, p.Lq - 1);
        const bf16_t* qp = p.q + b * p.q_sb + t * p.q_st + h * p.q_sh + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[nq][ks] = *(const bf16x8*)(qp + ks * 32);
    }
    // retire the Q loads here: left to the compiler, their s_waitcnt vmcnt lands on the first MFMA INSIDE the tile loop and
    // drains the K/V prefetch DMA on every iteration
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[nq][ks]));
    int q_abs[NQ];                           // absolute position of this lane's queries
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) q_abs[nq] = q0 + nq * 16 + c + p.q_offset;
    int64_t drop_row[NQ];                    // DROP: row of this lane's queries in the [B H Lq, S] score matrix
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) drop_row[nq] = ((int64_t)b * p.H + h) * p.Lq + min(q0 + nq * 16 + c, p.Lq - 1);
    const float* relrow[NQ];
    float gate[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) { relrow[nq] = nullptr; gate[nq] = 0.f; }
    if (REL && p.rel_table) {
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
            const int tq_ = min(q0 + nq * 16 + c, p.Lq - 1);
            relrow[nq] = p.rel_table + (int64_t)h * p.rel_stride + (p.rel_off - tq_);
            gate[nq] = (p.q_gate ? p.q_gate[((int64_t)b * p.H + h) * p.Lq + tq_] : 1.0f) * 1.4426950408889634f;
        }
    }

    f32x4 oacc[NQ][DB];
    float m_run[NQ], l_run[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
#pragma unroll
        for (int i = 0; i < DB; ++i) oacc[nq][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        m_run[nq] = NEG_BIG; l_run[nq] = 0.f;
    }

    // number of key tiles this workgroup needs
    int last_key = kvlen;                    // exclusive
    if (p.causal) last_key = min(last_key, qblk * QB + QB - 1 + p.q_offset + 1);
    const int ntiles = (last_key + 63) / 64;

    const bf16_t* kbase = p.k + b * p.k_sb + hk * p.k_sh;
    const bf16_t* vbase = p.v + b * p.v_sb + hk * p.v_sh;

    // K/V tiles arrive by LDS-DMA into a double buffer: the DMA of tile kt+1 is in flight while tile kt is consumed.
    // The LDS image is lane-linear per wave instruction (RPI rows x ROWB bytes); the XOR swizzle of K and V is applied to the SOURCE chunk.
    const int srow = lane / CH, sch = lane % CH;         // row inside the instruction's row group, 16-byte chunk
    // per-lane byte offsets inside a tile (loop invariant); tiles that reach past the last allocated row S-1 (only the final tile of a
    // buffer whose row count is not a multiple of 64, e.g. CLIP's 577 tokens) take the clamped 64-bit-address path instead
    uint32_t koff[NDMA], voff[NDMA];
    const bool off32_ok = 64LL * p.k_st * 2 + ROWB < (1LL << 31) && 64LL * p.v_st * 2 + ROWB < (1LL << 31);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int row = wave * WROWS + i * RPI + srow;
        int sw, swv;
        if (CH == 16) { sw = sch ^ (row & 15); swv = sch ^ ((row & 7) << 1); } else { sw = sch ^ ((row >> 1) & 7); swv = sch ^ (((row >> 1) & 3) << 1); }
        koff[i] = (uint32_t)(((int64_t)row * p.k_st + sw * 8) * 2);
        voff[i] = (uint32_t)(((int64_t)row * p.v_st + swv * 8) * 2);
    }
    auto stage = [&](int kt, int buf) {
        char* kb_ = lds + buf * (2 * TILE);
        char* vb_ = kb_ + TILE;
        if (!CL || (off32_ok && kt * 64 + 63 < p.S)) {          // CL = false: the launcher guarantees S % 64 == 0 and 32-bit tile offsets: the clamped path is compiled out
            const bf16_t* kt_k = kbase + (int64_t)kt * 64 * p.k_st;
            const bf16_t* kt_v = vbase + (int64_t)kt * 64 * p.v_st;
            const uint32_t l0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + (uint32_t)(buf * (2 * TILE));
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                dma16si(kt_k, koff[i], l0 + (uint32_t)((wave * WROWS + i * RPI) * ROWB));
                dma16si(kt_v, voff[i], l0 + (uint32_t)(TILE + (wave * WROWS + i * RPI) * ROWB));
            }
            return;
        }
        // rare path (the last tile of a buffer whose row count is not a multiple of 64).  The opaque copy of kt keeps hipcc from hoisting this
        // path's vector address arithmetic (~50 VALU instructions per operand) above the branch, where every tile of every launch paid for it
        int kt_c = kt;
        asm volatile("" : "+s"(kt_c));
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int row = wave * WROWS + i * RPI + srow;
            const int key = min(kt_c * 64 + row, p.S - 1);
            int sw, swv;
            if (CH == 16) { sw = sch ^ (row & 15); swv = sch ^ ((row & 7) << 1); } else { sw = sch ^ ((row >> 1) & 7); swv = sch ^ (((row >> 1) & 3) << 1); }
            dma16(kbase + (int64_t)key * p.k_st + sw * 8, kb_ + (wave * WROWS + i * RPI) * ROWB);
            dma16(vbase + (int64_t)key * p.v_st + swv * 8, vb_ + (wave * WROWS + i * RPI) * ROWB);
        }
    };
    if (ntiles > 0) stage(0, 0);

    // (a three-deep ring with two tiles in flight behind a counted vmcnt was measured and rejected: -6 %, the wait at the top of a tile is
    // not DMA latency)
    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        const char* kl = lds + buf * (2 * TILE);
        const char* vl = kl + TILE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt have landed
        __builtin_amdgcn_s_barrier();                          // ... and everyone's; all reads of the other buffer (tile kt-1) are done
        // causal: a tile that starts after this wave's last query contributes nothing (only the upper waves of a block reach
        // the block's last key tile); the wave still takes part in the staging and the barrier above
        const bool idle_tile = p.causal && kt * 64 > q0 + WQ - 1 + p.q_offset;
        if ((LATE == 0 || idle_tile) && kt + 1 < ntiles) stage(kt + 1, buf ^ 1);
        if (idle_tile) continue;

        // ---- S^T[key][query] = K · Q^T : every K fragment feeds the MFMAs of all NQ query blocks
        // K fragments are fetched one key block (KS reads) AHEAD of the MFMAs that use them, into their own registers
        f32x4 s[NQ][4];
        bf16x8 kfr[2][KS];
        auto ldk = [&](int kb, bf16x8 (&dst)[KS]) {
            const int row = kb * 16 + c;     // A operand row = key
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int ch = ks * 4 + g;
                int sw;
                if (CH == 16) sw = ch ^ (row & 15); else sw = ch ^ ((row >> 1) & 7);
                dst[ks] = *(const bf16x8*)(kl + row * ROWB + sw * 16);
            }
        };
        ldk(0, kfr[0]);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb < 3) ldk(kb + 1, kfr[(kb + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) s[nq][kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int nq = 0; nq < NQ; ++nq) s[nq][kb] = mc_mfma_16x16x32(kfr[kb & 1][ks], qf[nq][ks], s[nq][kb]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (LATE == 1 && kt + 1 < ntiles) stage(kt + 1, buf ^ 1);        // the next tile's DMA issued behind K.Q^T instead of beside its fragment reads
        // ---- mask, online softmax (lane owns query column c of each block; keys 16kb + 4g + r).  Only the diagonal tile (causal) and a
        // partial last tile need the per-score index tests; every other tile takes the mask-free path (wave-uniform branch per block).
        // (s_setprio(1) around the two MFMA clusters was measured: -3 %)
        // The running maximum is only raised (and the accumulators rescaled) when some query of the wave's block gained more than 2^RESC
        // over it; otherwise P is formed against the old maximum (values up to 2^RESC, exact in bf16's range) and the 8*DB accumulator
        // multiplies are skipped - the usual case after the first few tiles.  l and O always see the same factor.
        constexpr float RESC = 8.0f;
        bf16x8 pf[NQ][2];
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
            const bool need_mask = (kt * 64 + 63 >= kvlen) || (p.causal && kt * 64 + 63 > q0 + nq * 16 + p.q_offset) || (KM && p.key_valid != nullptr);
            const bool masked = need_mask || REL;
            float tmax = NEG_BIG;
            float lsum = 0.f;
            bool valid[4][4];
            // (1) tile maximum per query.  masked: scores scaled (and biased) in place, invalid keys left out; mask-free: the raw scores -
            // the scale is positive, so the maximum commutes with it and exp2(s*c - m) is one FMA per score
            if (masked) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 64 + kb * 16 + g * 4 + r;
                        bool ok = key < kvlen && (!p.causal || key <= q_abs[nq]);
                        if (KM && p.key_valid && ok) ok = p.key_valid[(int64_t)b * p.key_valid_sb + key] != 0;      // KM = false: compiled out (the LLM's unmasked launches)
                        valid[kb][r] = ok;
                        float sv = s[nq][kb][r] * p.scale_log2e;
                        if (REL && relrow[nq] && ok) sv += gate[nq] * relrow[nq][key];
                        s[nq][kb][r] = sv;
                        tmax = ok ? fmaxf(tmax, sv) : tmax;
                    }
                tmax = rows4_max(tmax);
            } else {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[nq][kb][r]);
                tmax = rows4_max(tmax);
                tmax *= p.scale_log2e;
            }
            // (2) the rare rescale, ONE copy of it for both paths, multiplying in place (round 3): with one copy per path the accumulators
            // came out of the masked and the mask-free path in different registers and the COMMON path paid 16 v_mov_b64 per query block
            // and tile to merge them (hipcc, ROCm 7.2)
            if (__builtin_amdgcn_ballot_w64(tmax > m_run[nq] + RESC) != 0) {
                const float m_new = fmaxf(m_run[nq], tmax);
                const float alpha = fast_exp2(m_run[nq] - m_new);
                m_run[nq] = m_new;
                l_run[nq] *= alpha;
#pragma unroll
                for (int i = 0; i < DB; ++i) scale_inplace(oacc[nq][i], alpha);
            }
            // (3) probabilities against the (possibly raised) running maximum
            if (masked) {
                const float mm = m_run[nq];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    float dm[4] = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (DROP) attn_dropout_quad(p.drop, drop_row[nq], p.S, kt * 64 + kb * 16 + g * 4, dm);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = valid[kb][r] ? fast_exp2(s[nq][kb][r] - mm) : 0.f;
                        lsum += pv;
                        pf[nq][kb >> 1][(kb & 1) * 4 + r] = (bf16_t)(DROP ? pv * dm[r] : pv);
                    }
                }
            } else {
                const float nm = -m_run[nq];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    float dm[4] = {1.f, 1.f, 1.f, 1.f};
                    if constexpr (DROP) attn_dropout_quad(p.drop, drop_row[nq], p.S, kt * 64 + kb * 16 + g * 4, dm);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = fast_exp2(fmaf(s[nq][kb][r], p.scale_log2e, nm));
                        lsum += pv;
                        pf[nq][kb >> 1][(kb & 1) * 4 + r] = (bf16_t)(DROP ? pv * dm[r] : pv);
                    }
                }
            }
            l_run[nq] += lsum;
        }

        // ---- O^T[d][query] += V^T · P^T ; V fragments through the transposing LDS read, each feeding all NQ query blocks
        const int tq = (lane & 15) >> 2, tp = lane & 3;
        // fragment f = pr * DB + db (key pair pr, d block db), fetched in groups of GV = 2 one group ahead of their MFMAs (groups of 4: -0.2 ... -1.1 % against 2; of 8: -1 ... -2 %; a three-deep ring of groups of 2: -0.3 %) (left to hipcc a
        // fragment's two transposing reads are issued one fragment - 32 MFMA cycles - ahead of its use, under the LDS latency); the MFMA
        // order per accumulator is unchanged (pr = 0 then pr = 1)
        constexpr int GV = 2, NG = 2 * DB / GV;
        bf16x8 vfr[2][GV];
        auto ldv = [&](int grp, bf16x8 (&dst)[GV]) {
#pragma unroll
            for (int i = 0; i < GV; ++i) {
                const int f = grp * GV + i, pr = f / DB, db = f % DB;
                const int key_lo = (2 * pr) * 16 + g * 4 + tq;
                const int key_hi = (2 * pr + 1) * 16 + g * 4 + tq;
                // 8 bytes at logical column (db*16 + tp*4) of row key: 16-byte chunk db*2 + (tp>>1), XOR-swizzled per row (256-byte rows = one
                // full bank line each).  A transposing read is served in two groups of 32 lanes = 8 rows x 2 chunks x 2 halves: with
                // chunk ^ (2 * (row & 7)) the 16 (row, chunk) pairs of a group land on 16 different chunk slots = all 64 banks once
                // (chunk ^ (row & 15), K's swizzle, puts rows k and k^1 on the same slot: 2-way conflicts, 30 % of the LDS cycles measured)
                const int chv = db * 2 + (tp >> 1);
                // (128-byte rows, D = 64: two rows per bank line, 8 rows x 2 chunks of a group -> chunk ^ (2 * ((row >> 1) & 3)))
                const int sw_lo = (CH == 16) ? (chv ^ ((key_lo & 7) << 1)) : (chv ^ (((key_lo >> 1) & 3) << 1));
                const int sw_hi = (CH == 16) ? (chv ^ ((key_hi & 7) << 1)) : (chv ^ (((key_hi >> 1) & 3) << 1));
                const bf16x4 lo = mc_ds_read_tr16(
                    (__attribute__((address_space(3))) bf16x4*)(char*)(vl + key_lo * ROWB + sw_lo * 16 + (tp & 1) * 8));
                const bf16x4 hi = mc_ds_read_tr16(
                    (__attribute__((address_space(3))) bf16x4*)(char*)(vl + key_hi * ROWB + sw_hi * 16 + (tp & 1) * 8));
                bf16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                dst[i] = vf;
            }
        };
        ldv(0, vfr[0]);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            if (grp + 1 < NG) ldv(grp + 1, vfr[(grp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GV; ++i) {
                const int f = grp * GV + i, pr = f / DB, db = f % DB;
#pragma unroll
                for (int nq = 0; nq < NQ; ++nq) oacc[nq][db] = mc_mfma_16x16x32(vfr[grp & 1][i], pf[nq][pr], oacc[nq][db]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- finalize: total row sum over the 4 lane groups that share query c
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        const float l = rows4_sum(l_run[nq]);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int t = q0 + nq * 16 + c;
        if (p.lse && t < p.Lq && g == 0) p.lse[((int64_t)b * p.H + h) * p.Lq + t] = l > 0.f ? m_run[nq] + log2f(l) : NEG_BIG;
        if (t < p.Lq) {
            const int64_t row = p.out_map ? p.out_map[b * p.Lq + t] : (int64_t)b * p.Lq + t;
            if (row >= 0) {
                bf16_t* op = p.o + row * p.o_row_stride + h * D;
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    bf16x4 ov = {(bf16_t)(oacc[nq][db][0] * inv), (bf16_t)(oacc[nq][db][1] * inv), (bf16_t)(oacc[nq][db][2] * inv),
                                 (bf16_t)(oacc[nq][db][3] * inv)};
                    *(bf16x4*)(op + db * 16 + g * 4) = ov;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Prefill attention on the 32x32x16 MFMA (round 5) - the LLM's launches: head_dim 128, no relative-position bias, no dropout, no per-key mask.
// ------------------------------------------------------------------------------------------
// The kernel above is ISSUE-bound (round-4 counters: each wave issuing 44 % of its cycles, the matrix pipe 41.6 % busy): per 64-key tile and
// wave it issues 64 v_mfma_f32_16x16x32 - each holds the SIMD's vector issue port for 8 of its 16 cycles - beside ~250 softmax VALU
// instructions and 48 LDS reads.  The 32x32x16 shape does the same FLOPs in 32 instructions that hold the port for 8 of their 32 cycles:
// 768 free issue cycles per tile for the same softmax.  Same algorithm, other fragment maps (cdna_hip_programming.md §3):
//   S^T[key][q] = K . Q^T   A = K rows (lane r = key, half h: d = 16 ks + 8 h + j), B = Q^T (lane r = query, same d); the 32 x 32 result has the
//                           QUERY on the lane (r) and 16 keys (reg & 3) + 8 (reg >> 2) + 4 h in the registers: row statistics are in-lane + ONE
//                           v_permlane32_swap across the halves.
//   O^T[d][q] += V^T . P^T  P^T is the accumulator tile used as the next B operand: registers 8 s .. 8 s + 7 -> bf16 = the fragment of k-step s,
//                           whose element j of half h is key 16 s + 8 (j >> 2) + 4 h + (j & 3); V^T (A operand, lane r = d) takes the same keys:
//                           two transposing reads of 4 keys each (ds_read_b64_tr_b16: rows 16 s + 4 h .. + 3 and + 8).  O^T keeps the query on
//                           the lane, so the (rare) rescale and the final 1 / l are per-lane multiplies.
// One workgroup = 4 waves x 32 queries; K / V tiles of 64 keys, double-buffered by LDS-DMA, both XOR-swizzled by
// chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)) (conflict-free for the row reads of K and the transposing reads of V of this shape).
// The online softmax (deferred maximum, P rounded to the storage type before P.V, fp32 row sums) is the one of attn_prefill_kernel: results
// agree with it to fp32 summation order (the P.V sums run over the keys in another order); tests bound both against the fp32 reference.
template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_prefill32_kernel(AttnParams p) {
    constexpr int D = 128, ROWB = 256, TILE = 64 * ROWB, KS = D / 16, DB = D / 32, QB = 128, WQ = 32;
    extern __shared__ __attribute__((aligned(16))) char lds[];               // [buffer][K | V] = 4 x 16 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int qblk, hd, b;
    {   // XCD-aware block order (see attn_prefill_kernel)
        const int nx = gridDim.x, ny = gridDim.y;
        const int total = nx * ny * (int)gridDim.z;
        const int lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
        const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7, idx = lin >> 3;
        const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        qblk = nx - 1 - v % nx;
        hd = (v / nx) % ny;
        b = v / (nx * ny);
    }
    const int hk = hd / (p.H / p.Hkv);
    const int q0 = qblk * QB + wave * WQ;
    const int kvlen = p.kv_lens ? min(p.kv_lens[b], p.S) : p.S;
    const int tq_row = min(q0 + r, p.Lq - 1);
    const int q_abs = q0 + r + p.q_offset;

    bf16x8 qf[KS];
    {
        const bf16_t* qp = p.q + b * p.q_sb + (int64_t)tq_row * p.q_st + hd * p.q_sh + h * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks]));          // retired here, not on the first MFMA inside the loop
    }
    f32x16 oacc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[i][e] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    int last_key = kvlen;
    if (CAUSAL) last_key = min(last_key, qblk * QB + QB - 1 + p.q_offset + 1);
    const int ntiles = (last_key + 63) / 64;
    const bf16_t* kbase = p.k + b * p.k_sb + hk * p.k_sh;
    const bf16_t* vbase = p.v + b * p.v_sb + hk * p.v_sh;

    // staging: 4 rows x 256 B per 1-KiB DMA instruction, wave w stages rows 16 w .. 16 w + 15 of K and of V (4 + 4 instructions)
    const int srow = lane >> 4, sch = lane & 15;
    uint32_t koff[4], voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 16 + i * 4 + srow;
        const int sw = sch ^ (((row & 3) << 2) | ((row >> 2) & 3));
        koff[i] = (uint32_t)(((int64_t)row * p.k_st + sw * 8) * 2);
        voff[i] = (uint32_t)(((int64_t)row * p.v_st + sw * 8) * 2);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    auto stage = [&](int kt, int buf) {
        const bf16_t* kt_k = kbase + (int64_t)kt * 64 * p.k_st;
        const bf16_t* kt_v = vbase + (int64_t)kt * 64 * p.v_st;
        const uint32_t l0 = lds0 + (uint32_t)(buf * (2 * TILE));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dma16si(kt_k, koff[i], l0 + (uint32_t)((wave * 16 + i * 4) * ROWB));
            dma16si(kt_v, voff[i], l0 + (uint32_t)(TILE + (wave * 16 + i * 4) * ROWB));
        }
    };
    if (ntiles > 0) stage(0, 0);

    // loop-invariant LDS offsets.  K row read: row = 32 kb + r, chunk 2 ks + h.  V transposing read: 16-lane group gi = lane >> 4 covers the
    // d columns 16 (gi & 1) .. + 15 of a 32-wide d block; its lane 4 q + p supplies row (key base + q), columns 4 p .. 4 p + 3
    const int li = lane & 15, tq = li >> 2, tp = li & 3, dhalf = (lane >> 4) & 1;
    const float scale = p.scale_log2e;

    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        const char* kl = lds + buf * (2 * TILE);
        const char* vl = kl + TILE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // causal: a tile that starts after this wave's last query contributes nothing (only the upper waves of a block reach the block's last
        // key tile); the wave still takes part in the staging (ONE staging site: a second copy of its eight DMA instructions and their
        // address registers pushed the kernel over the register limit) and in the barrier
        const bool idle_tile = CAUSAL && kt * 64 > q0 + WQ - 1 + p.q_offset;
        // ---- S^T = K . Q^T : two 32-key blocks, 8 k-steps of 16 over the head dim; K fragments fetched four k-steps ahead
        f32x16 s[2];
        bf16x8 kfr[2][4];
        if (!idle_tile) {
        auto ldk = [&](int grp, bf16x8 (&dst)[4]) {           // grp = 2 kb + (ks >> 2)
            const int kb = grp >> 1, ks0 = (grp & 1) * 4;
            const int row = kb * 32 + r;
            const int f = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i] = *(const bf16x8*)(kl + row * ROWB + (((ks0 + i) * 2 + h) ^ f) * 16);
        };
        ldk(0, kfr[0]);
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            if (grp < 3) ldk(grp + 1, kfr[(grp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if ((grp & 1) == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[grp >> 1][e] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s[grp >> 1] = mc_mfma_32x32x16(kfr[grp & 1][i], qf[(grp & 1) * 4 + i], s[grp >> 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (kt + 1 < ntiles) stage(kt + 1, buf ^ 1);          // behind K.Q^T, in front of the softmax's VALU-only stretch
        if (idle_tile) continue;
        // ---- online softmax: this lane's query against keys 64 kt + 32 kb + (e & 3) + 8 (e >> 2) + 4 h
        const bool need_mask = (kt * 64 + 63 >= kvlen) || (CAUSAL && kt * 64 + 63 > q0 + p.q_offset);
        constexpr float RESC = 8.0f;
        float tmax = NEG_BIG;
        if (need_mask) {                                       // masked scores are replaced in place by -3e38: they lose every maximum and exp2 to 0
            // key = 64 kt + 4 h + c with the compile-time c = 32 kb + (e & 3) + 8 (e >> 2): ONE per-lane limit, compared with constants (the
            // thresholds must not become 32 hoisted registers: the kernel sits at the register limit of two waves per SIMD)
            int lim = kvlen - 1 - kt * 64 - 4 * h;
            if (CAUSAL) lim = min(lim, q_abs - kt * 64 - 4 * h);
            asm volatile("" : "+v"(lim));
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[kb][e] = (kb * 32 + (e & 3) + 8 * (e >> 2)) <= lim ? s[kb][e] : -3.0e38f;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int e = 0; e < 16; ++e) tmax = fmaxf(tmax, s[kb][e]);
        {
            auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
            tmax = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));
        }
        tmax = tmax > NEG_BIG ? tmax * scale : NEG_BIG;       // the scale is positive: the maximum commutes with it (no valid key yet: stays NEG_BIG)
        if (__builtin_amdgcn_ballot_w64(tmax > m_run + RESC) != 0) {
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = fast_exp2(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[i][e] *= alpha;
        }
        bf16x8 pf[2][2];
        {
            const float nm = -m_run;
            float lsum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pv = fast_exp2(fmaf(s[kb][e], scale, nm));
                    lsum += pv;
                    pf[kb][e >> 3][e & 7] = (bf16_t)pv;
                }
            l_run += lsum;
        }
        // ---- O^T += V^T . P^T : per (key block kb, k-step st of 16 keys) and d block db one fragment = two transposing reads
        bf16x8 vfr[2][2];
        auto ldv = [&](int grp, bf16x8 (&dst)[2]) {           // grp = (kb * 2 + st) * 2 + (db >> 1): fragments db = 2 (grp & 1), + 1
            const int ksx = grp >> 1;                          // kb * 2 + st
            const int key0 = (ksx >> 1) * 32 + (ksx & 1) * 16 + 4 * h + tq;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int db = (grp & 1) * 2 + i;
                const int ch = db * 4 + dhalf * 2 + (tp >> 1);
                const int ka = key0, kb_ = key0 + 8;
                const int fa = ((ka & 3) << 2) | ((ka >> 2) & 3), fb = ((kb_ & 3) << 2) | ((kb_ >> 2) & 3);
                const bf16x4 lo = mc_ds_read_tr16((mc_lds_void*)(vl + ka * ROWB + ((ch ^ fa) * 16) + (tp & 1) * 8));
                const bf16x4 hi = mc_ds_read_tr16((mc_lds_void*)(vl + kb_ * ROWB + ((ch ^ fb) * 16) + (tp & 1) * 8));
                bf16x8 vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                dst[i] = vf;
            }
        };
        ldv(0, vfr[0]);
#pragma unroll
        for (int grp = 0; grp < 8; ++grp) {
            if (grp < 7) ldv(grp + 1, vfr[(grp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int ksx = grp >> 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int db = (grp & 1) * 2 + i;
                oacc[db] = mc_mfma_32x32x16(vfr[grp & 1][i], pf[ksx >> 1][ksx & 1], oacc[db]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- finalize
    {
        auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        const float l = __uint_as_float(sw_[0]) + __uint_as_float(sw_[1]);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int t = q0 + r;
        if (p.lse && t < p.Lq && h == 0) p.lse[((int64_t)b * p.H + hd) * p.Lq + t] = l > 0.f ? m_run + log2f(l) : NEG_BIG;
        int64_t row = -1;
        if (t < p.Lq) row = p.out_map ? p.out_map[b * p.Lq + t] : (int64_t)b * p.Lq + t;
        bf16_t* op = p.o + (row < 0 ? 0 : row) * p.o_row_stride + hd * D;
        // oacc[db][e]: d = 32 db + 8 (e >> 2) + 4 h + (e & 3).  v_permlane32_swap pairs the d groups 2 g, 2 g + 1 of the two halves so that the
        // lower half stores the 8 consecutive d of group 2 g and the upper half those of group 2 g + 1: 16 bytes per lane and store
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int gq = 0; gq < 4; gq += 2) {
                bf16x4 a4, b4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { a4[e] = (bf16_t)(oacc[db][gq * 4 + e] * inv); b4[e] = (bf16_t)(oacc[db][(gq + 1) * 4 + e] * inv); }
                const u32x2 pa = __builtin_bit_cast(u32x2, a4), pb = __builtin_bit_cast(u32x2, b4);
                auto r0 = __builtin_amdgcn_permlane32_swap(pa[0], pb[0], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(pa[1], pb[1], false, false);
                // lower half: r0[0], r1[0] = own group gq (d + 0 .. 3), r0[1], r1[1] = the upper half's group gq (d + 4 .. 7)
                // upper half: r0[0], r1[0] = the lower half's group gq + 1 (d + 0 .. 3), r0[1], r1[1] = own group gq + 1 (d + 4 .. 7)
                const u32x4 ov = {r0[0], r1[0], r0[1], r1[1]};
                if (row >= 0) *(u32x4*)(op + db * 32 + (gq + h) * 8) = ov;
            }
    }
}
// ------------------------------------------------------------------------------------------
// tiny sequences (Lq, S <= 8): the temporal attention of LanguageBind-Video (languagebind/video/modeling_video.py:105-130: every patch
// position attends over its t = 8 frames - 4112 sequences x 16 heads of 8 x 8 scores per clip batch).  The flash kernel above spends a
// 64 x 64 MFMA tile and two 64-row LDS stages on each of them (4 TFLOP/s, 260 us per layer); this is a memory-bound problem: q, k, v read
// once, o written once.  One thread = (sequence, head, query, half of head_dim): its 8 scores are dot products over its half, summed with
// the partner lane, fp32 softmax, then its half of the output row.
template <int D>
__global__ __launch_bounds__(256) void attn_tiny_kernel(AttnParams p) {
    constexpr int HALF = D / 2, NV = HALF / 8;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int half = (int)(gid & 1);
    const int t = (int)((gid >> 1) & 7);
    const int64_t bh = gid >> 4;
    if (bh >= (int64_t)p.B * p.H) return;
    const int b = (int)(bh / p.H), h = (int)(bh % p.H);
    const int hk = h / (p.H / p.Hkv);
    const int kvlen = p.kv_lens ? min(p.kv_lens[b], p.S) : p.S;
    const int tq = min(t, p.Lq - 1);
    // batch offsets: one stride, or the two-level form
    const int64_t bq = p.b_inner ? (int64_t)(b / p.b_inner) * p.q_sb + (int64_t)(b % p.b_inner) * p.sbi : (int64_t)b * p.q_sb;
    const int64_t bk = p.b_inner ? (int64_t)(b / p.b_inner) * p.k_sb + (int64_t)(b % p.b_inner) * p.sbi : (int64_t)b * p.k_sb;
    const int64_t bv = p.b_inner ? (int64_t)(b / p.b_inner) * p.v_sb + (int64_t)(b % p.b_inner) * p.sbi : (int64_t)b * p.v_sb;
    const bf16_t* qp = p.q + bq + (int64_t)tq * p.q_st + h * p.q_sh + half * HALF;
    float qv[HALF];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bf16x8 x = *(const bf16x8*)(qp + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[i * 8 + j] = (float)x[j];
    }
    float sc[8];
    float m = NEG_BIG;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int key = min(j, p.S - 1);
        const bf16_t* kp = p.k + bk + (int64_t)key * p.k_st + hk * p.k_sh + half * HALF;
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bf16x8 x = *(const bf16x8*)(kp + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf(qv[i * 8 + e], (float)x[e], d);
        }
        d += __shfl_xor(d, 1, 64);                          // the other half of head_dim
        const bool ok = j < kvlen && (!p.causal || j <= tq + p.q_offset);
        sc[j] = ok ? d * p.scale_log2e : NEG_BIG;
        m = fmaxf(m, sc[j]);
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = sc[j] > NEG_BIG * 0.5f ? fast_exp2(sc[j] - m) : 0.f; l += sc[j]; }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    float o[HALF];
#pragma unroll
    for (int i = 0; i < HALF; ++i) o[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int key = min(j, p.S - 1);
        const bf16_t* vp = p.v + bv + (int64_t)key * p.v_st + hk * p.v_sh + half * HALF;
        // P is rounded to bf16 before it multiplies V, as in the flash kernel (and normalised by the fp32 sum)
        const float pj = (float)(bf16_t)sc[j];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bf16x8 x = *(const bf16x8*)(vp + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[i * 8 + e] = fmaf(pj, (float)x[e], o[i * 8 + e]);
        }
    }
    if (t < p.Lq) {
        const int64_t row = p.out_map ? p.out_map[b * p.Lq + t] : (int64_t)b * p.Lq + t;
        if (row >= 0) {
            bf16_t* op = p.o + row * p.o_row_stride + h * D + half * HALF;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                bf16x8 x;
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (bf16_t)(o[i * 8 + e] * inv);
                *(bf16x8*)(op + i * 8) = x;
            }
        }
        if (p.lse && half == 0) p.lse[((int64_t)b * p.H + h) * p.Lq + t] = l > 0.f ? m + log2f(l) : NEG_BIG;
    }
}

// ------------------------------------------------------------------------------------------
// decode: Lq == 1
// ------------------------------------------------------------------------------------------
struct DecodeParams {
    const bf16_t* q; int64_t q_sb, q_sh;
    const bf16_t* k; int64_t k_sb, k_st, k_sh;
    const bf16_t* v; int64_t v_sb, v_st, v_sh;
    bf16_t* o; int64_t o_sb;                       // out[b*o_sb + h*D + d]
    float* ws;                                     // [B*H, ws_chunks, D + 2] partials (m, l, acc) when the workgroup does not combine itself
    const int32_t* kv_lens;
    int B, H, Hkv, S, nsplit;
    float scale_log2e;
    // fused RoPE + KV append (decode step of LocalLoraAttention.forward, multimodal_llama.py:281-289): when qkv is set, q / k_new / v_new
    // are row b of qkv [B, (H + 2 Hkv) * D] (pre-rotary); the key at position kv_lens[b]-1 is this token: it is rotated in registers,
    // attended from registers and appended to the caches by this kernel (MHA only: H == Hkv)
    const bf16_t* qkv; int64_t qkv_ld;
    const float* cosT; const float* sinT;
    bf16_t* k_out; bf16_t* v_out;
    const uint8_t* key_valid; int64_t key_valid_sb;          // optional per-key validity of the CACHED keys (see AttnParams)
    int ws_chunks, ws_lds;                                   // chunk partials per (b, h) in ws; ws_lds: the workgroup keeps them in LDS and combines itself
};

// Round 6: batch-invariant schedule.  The cached keys 0 .. cache_len - 1 of a sequence are cut into CHUNKS of kDecChunk keys - a function of
// the sequence's own length only.  One workgroup computes one chunk's partial (m, l, acc[D]) from a fresh state, always the same way: its 4
// waves take the chunk's keys in groups of 4 (wave w: keys = 4 w + slot mod 16; a workgroup iteration reads 64 consecutive keys = 16 KiB of K
// and of V, whole DRAM pages), every (wave, slot) runs the online softmax over its keys in ascending order, the 4 slots of a wave are merged
// by the xor tree, the 4 waves through LDS in wave order.  The partials of a (b, h) are combined in chunk order by decode_combine() - by the
// workgroup itself from LDS when it owns every chunk (gridDim.y == 1), else by attn_decode_combine_kernel from the workspace.  Which
// workgroup computes which chunk (gridDim.y, i.e. `nsplit`, chosen from the batch size to fill the chip) changes neither a partial nor
// the combine order: the output row of a sequence is the same bits at any B, any nsplit and any Smax.  The token's own key (fused mode:
// it comes from registers, not from the cache) is one more partial, after the chunks of the cached keys: (m, l, acc) = (q.k, 1, v).
// (Chunks owned by single waves - 128 or 256 keys each, no barrier - measured 5-6 % slower at B = 48: a wave's 4 KiB requests open four
// times the DRAM pages of the workgroup's 16 KiB ones; profiles/r06_probes/decode_attn_variants.log.)
constexpr int kDecChunk = 512;
constexpr int kDecLdsChunks = 40;            // partials a workgroup holds in LDS (Smax <= 19 968; 21 KiB): above that the workspace route runs

template <int D>
__device__ __forceinline__ float decode_combine(const float* parts, int n, int d) {
    float mm = NEG_BIG;
    for (int c = 0; c < n; ++c) mm = fmaxf(mm, parts[c * (D + 2) + D]);
    float ll = 0.f, aa = 0.f;
    for (int c = 0; c < n; ++c) {
        const float a = fast_exp2(parts[c * (D + 2) + D] - mm);
        ll = fmaf(parts[c * (D + 2) + D + 1], a, ll);
        aa = fmaf(parts[c * (D + 2) + d], a, aa);
    }
    return ll > 0.f ? aa / ll : 0.f;
}

// UN = key groups per wave and request (4: 8 KiB in flight per wave at 4 waves per SIMD - the batch-48 shape; 8: 16 KiB, for launches too
// small to fill the chip with waves - batch 1: 6 workgroups per head).  A (wave, slot) sees the same keys in the same order either way.
// (UN = 16 without the second register buffer - half a chunk per request, for grids of at most one workgroup per CU - measured no better:
// batch 1 19.5 vs 18.6 us, batch 8 at nsplit 1 81.7 vs 73.9; not in the tree.)
template <int D, int UN>
__global__ __launch_bounds__(256, UN == 4 ? 4 : 2) void attn_decode_kernel(DecodeParams p) {
    constexpr int LPK = D / 8;               // lanes per key
    constexpr int KPW = 64 / LPK;            // keys per wave load
    constexpr int KPI = 4 * KPW * UN;        // keys per workgroup iteration
    __shared__ float red[4][D + 2];
    extern __shared__ float parts_lds[];     // [chunks][D + 2] when the workgroup combines itself (one workgroup per head, chunks fit)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H;
    const int hk = h / (p.H / p.Hkv);
    const int kvlen = p.kv_lens ? min(p.kv_lens[b], p.S) : p.S;
    const int slot = lane / LPK, dl = (lane % LPK) * 8;
    const bool fused = p.qkv != nullptr;
    const int cache_len = fused ? kvlen - 1 : kvlen;          // keys read from the cache (fused: the new token's key comes from registers)
    const int nch = (cache_len + kDecChunk - 1) / kDecChunk;       // chunks of cached keys; the fused token's partial is number nch
    const bool in_lds = p.ws_lds != 0;
    // partial (chunk c, entry j): LDS or the workspace - two address spaces, selected at the store (one generic pointer held across the
    // key loop cost two registers the 128-register instantiation does not have)
    auto part_store = [&](int c_, int j_, float v_) {
        if (in_lds) parts_lds[c_ * (D + 2) + j_] = v_;
        else p.ws[((int64_t)bh * p.ws_chunks + c_) * (D + 2) + j_] = v_;
    };

    const bf16_t* kb = p.k + b * p.k_sb + hk * p.k_sh + dl;
    const bf16_t* vb = p.v + b * p.v_sb + hk * p.v_sh + dl;
    // The first batch of cached K/V rows is requested before anything else: the q/k rotation and the cache append below (a dependent
    // chain of small loads, and a store the compiler will not move loads across) then run under its HBM latency instead of ahead of it.
    bf16x8 k8[UN], v8[UN];
    auto request = [&](int j) {               // j: first key of this wave's share of the workgroup iteration
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kc = min(j + u * 4 * KPW + slot, p.S - 1);
            k8[u] = __builtin_nontemporal_load((const bf16x8*)(kb + (int64_t)kc * p.k_st));
            v8[u] = __builtin_nontemporal_load((const bf16x8*)(vb + (int64_t)kc * p.v_st));
        }
    };
    int c = blockIdx.y;
    if (c < nch) request(c * kDecChunk + wave * KPW);

    float qv[8];
    if (fused) {
        float knew[8];
        const int pos = kvlen - 1;
        const bf16_t* row = p.qkv + (int64_t)b * p.qkv_ld;
        const bf16x8 q8 = *(const bf16x8*)(row + h * D + dl);
        const bf16x8 kn8 = *(const bf16x8*)(row + (p.H + hk) * D + dl);
        const bf16x8 vn8 = *(const bf16x8*)(row + (p.H + p.Hkv + hk) * D + dl);
        // rotate-half: lanes dl < D/2 pair with lane + LPK/2 (dl + D/2)
        const bool lo = dl < D / 2;
        const float* cr = p.cosT + (int64_t)pos * (D / 2) + (lo ? dl : dl - D / 2);
        const float* sr = p.sinT + (int64_t)pos * (D / 2) + (lo ? dl : dl - D / 2);
        bf16x8 kr;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float qa = (float)q8[i], ka = (float)kn8[i];
            const float qb = __shfl_xor(qa, LPK / 2, 64), kb_ = __shfl_xor(ka, LPK / 2, 64);
            const float cs = cr[i], sn = lo ? -sr[i] : sr[i];
            // lo: x*c - partner*s ; hi: x*c + partner*s.  Rounded to bf16 like the stored q / cached k of the unfused path
            qv[i] = (float)(bf16_t)fmaf(qa, cs, (qb * sn)) * p.scale_log2e;      // = mc_rope_pair (common.h), bit for bit
            kr[i] = (bf16_t)fmaf(ka, cs, (kb_ * sn));
            knew[i] = (float)kr[i];
        }
        if (blockIdx.y == 0 && wave == 0 && slot == 0) {
            if (h % (p.H / p.Hkv) == 0) {
                *(bf16x8*)(p.k_out + b * p.k_sb + hk * p.k_sh + (int64_t)pos * p.k_st + dl) = kr;
                *(bf16x8*)(p.v_out + b * p.v_sb + hk * p.v_sh + (int64_t)pos * p.v_st + dl) = vn8;
            }
            // this token's own key / value: partial number nch
            float sdot = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sdot = fmaf(qv[i], knew[i], sdot);
#pragma unroll
            for (int o = 1; o < LPK; o <<= 1) sdot += __shfl_xor(sdot, o, 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) part_store(nch, dl + i, (float)vn8[i]);
            if (dl == 0) { part_store(nch, D, sdot); part_store(nch, D + 1, 1.f); }
        }
    } else {
        const bf16x8 q8 = *(const bf16x8*)(p.q + b * p.q_sb + h * p.q_sh + dl);
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[i] = (float)q8[i] * p.scale_log2e;
    }

    for (; c < nch; c += gridDim.y) {
        const int j1 = min(cache_len, (c + 1) * kDecChunk);
        const int cn = c + gridDim.y;
        float m = NEG_BIG, l = 0.f, acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        // register double buffer: the next batch (of this chunk, or the first of the workgroup's next one) is requested before the
        // current one is consumed
        for (int j = c * kDecChunk + wave * KPW; j < j1; j += KPI) {
            bf16x8 kc8[UN], vc8[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) { kc8[u] = k8[u]; vc8[u] = v8[u]; }
            if (j + KPI < j1) request(j + KPI);
            else if (cn < nch) request(cn * kDecChunk + wave * KPW);
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int key = j + u * 4 * KPW + slot;
                // (explicit fmaf everywhere: with -ffp-contract=fast the two instantiations of this loop - UN = 4 / 8 - were contracted
                // differently, a*b + c*d has two fused forms, and a row's bits depended on the launch size: caught by
                // test_fixture_rows_inside_the_benchmarked_48_row_batch)
                float sdot = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) sdot = fmaf(qv[i], (float)kc8[u][i], sdot);
#pragma unroll
                for (int o = 1; o < LPK; o <<= 1) sdot += __shfl_xor(sdot, o, 64);
                bool ok = key < j1;
                if (p.key_valid && ok) ok = p.key_valid[(int64_t)b * p.key_valid_sb + key] != 0;
                const float mn = ok ? fmaxf(m, sdot) : m;
                const float a = fast_exp2(m - mn);
                const float pv = ok ? fast_exp2(sdot - mn) : 0.f;
                m = mn;
                l = fmaf(l, a, pv);
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    const f32x2_ a2 = {acc[i], acc[i + 1]}, v2 = {(float)vc8[u][i], (float)vc8[u][i + 1]}, s2 = {a, a}, p2 = {pv, pv};
                    const f32x2_ r2 = __builtin_elementwise_fma(p2, v2, a2 * s2);
                    acc[i] = r2[0]; acc[i + 1] = r2[1];
                }
            }
        }
        // a wave whose share of the chunk is empty (a short last chunk) still has this chunk's request in flight: none was issued for it
        // (the loop's first request of a chunk is issued by the previous chunk's last iteration or the prologue - only when the wave has keys)
        // merge the KPW key slots of the wave (lanes with equal dl)
#pragma unroll
        for (int o = LPK; o < 64; o <<= 1) {
            const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
            const float mn = fmaxf(m, m2);
            const float a1 = fast_exp2(m - mn), a2 = fast_exp2(m2 - mn);
            l = fmaf(l2, a2, (l * a1));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float o2 = __shfl_xor(acc[i], o, 64);
                acc[i] = fmaf(o2, a2, (acc[i] * a1));
            }
            m = mn;
        }
        if (slot == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) red[wave][dl + i] = acc[i];
            if (dl == 0) { red[wave][D] = m; red[wave][D + 1] = l; }
        }
        __syncthreads();
        if (tid < D) {
            float mm = NEG_BIG;
#pragma unroll
            for (int w = 0; w < 4; ++w) mm = fmaxf(mm, red[w][D]);
            float ll = 0.f, aa = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float a = fast_exp2(red[w][D] - mm);
                ll = fmaf(red[w][D + 1], a, ll);
                aa = fmaf(red[w][tid], a, aa);
            }
            part_store(c, tid, aa);
            if (tid == 0) { part_store(c, D, mm); part_store(c, D + 1, ll); }
        }
        __syncthreads();
    }
    if (!in_lds) return;
    __syncthreads();
    if (tid < D) p.o[b * p.o_sb + h * D + tid] = (bf16_t)decode_combine<D>(parts_lds, nch + (fused ? 1 : 0), tid);
}

template <int D>
__global__ void attn_decode_combine_kernel(DecodeParams p) {
    const int bh = blockIdx.x, b = bh / p.H, h = bh % p.H, d = threadIdx.x;
    const int kvlen = p.kv_lens ? min(p.kv_lens[b], p.S) : p.S;
    const int fused = p.qkv != nullptr;
    const int nch = (kvlen - fused + kDecChunk - 1) / kDecChunk + fused;
    p.o[b * p.o_sb + h * D + d] = (bf16_t)decode_combine<D>(p.ws + (int64_t)bh * p.ws_chunks * (D + 2), nch, d);
}

// ------------------------------------------------------------------------------------------
// mc_attn_mask (include/mc_hip.h): the optional per-key validity bytes and the two-level batch index of a launch, as explicit arguments.
// (Rounds 2-5 carried them as thread-local "next launch" state; a struct argument cannot leak into a later launch or be left behind by a
// refused one.)
static void mask_fields(const mc_attn_mask* mk, const uint8_t*& kv, int64_t& kv_sb, int& b_inner, int64_t& sbi) {
    kv = mk ? (const uint8_t*)mk->key_valid : nullptr;
    kv_sb = mk && mk->key_valid ? mk->key_valid_stride : 0;
    b_inner = mk ? mk->b_inner : 0;
    sbi = mk ? mk->inner_stride : 0;
}

// Diagnostic word (kernel A/B variants): probes build only (csrc/Makefile `probes`)
#ifdef MC_PROBES
static int g_attn_dbg = 0;
extern "C" int mc_attn_debug(int v) { g_attn_dbg = v; return 0; }      // bit 0: force the 64-query kernel, bit 1: allow the 128-query one at any length (A/B timing)
#else
constexpr int g_attn_dbg = 0;
#endif

extern "C" int mc_attn_prefill_lse_bf16(const void* q, int64_t q_sb, int64_t q_st, int64_t q_sh, const void* k, int64_t k_sb,
                                        int64_t k_st, int64_t k_sh, const void* v, int64_t v_sb, int64_t v_st, int64_t v_sh,
                                        void* o, int64_t o_row_stride, const int32_t* out_map, const int32_t* kv_lens, int B,
                                        int H, int Hkv, int Lq, int S, int D, int causal, int q_offset, float scale,
                                        const float* rel_table, int rel_stride, int rel_off, const float* q_gate, float* lse,
                                        const mc_attn_mask* mask, void* stream) {
    const uint8_t* kv_once = nullptr; int64_t kv_once_sb = 0;
    int b_inner_once = 0; int64_t sbi_once = 0;
    mask_fields(mask, kv_once, kv_once_sb, b_inner_once, sbi_once);
    MC_CHECK_ARG(b_inner_once >= 0 && sbi_once % 8 == 0, "mc_attn_prefill_bf16: bad two-level batch index");
    MC_CHECK_ARG(q && k && v && o, "mc_attn_prefill_bf16: null pointer");
    MC_CHECK_ARG(D == 64 || D == 128, "mc_attn_prefill_bf16: head_dim %d not supported (64 or 128)", D);
    MC_CHECK_ARG(B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && Lq > 0 && S > 0, "mc_attn_prefill_bf16: bad shape");
    MC_CHECK_ARG((q_st % 8 | q_sh % 8 | k_st % 8 | k_sh % 8 | v_st % 8 | v_sh % 8 | q_sb % 8 | k_sb % 8 | v_sb % 8) == 0,
                 "mc_attn_prefill_bf16: strides must be multiples of 8 elements");
    AttnParams p{(const bf16_t*)q, q_sb, q_st, q_sh, (const bf16_t*)k, k_sb, k_st, k_sh, (const bf16_t*)v, v_sb, v_st, v_sh,
                 (bf16_t*)o, o_row_stride, out_map, kv_lens, B, H, Hkv, Lq, S, causal, q_offset,
                 scale * 1.4426950408889634f, rel_table, q_gate, rel_stride, rel_off, lse};
    p.key_valid = kv_once; p.key_valid_sb = kv_once_sb;
    p.b_inner = b_inner_once; p.sbi = sbi_once;
    MC_CHECK_ARG(p.b_inner == 0 || (Lq <= 8 && S <= 8 && !rel_table && !p.key_valid && B % p.b_inner == 0),
                 "mc_attn_prefill_bf16: a two-level batch index (mc_attn_mask.b_inner) is only defined for the Lq, S <= 8 kernel");
    hipStream_t s = (hipStream_t)stream;
    // Workgroup shape: QW waves x NQ 16-query blocks per wave.  D = 128 without the relative-position bias (the LLM prefill, training):
    // two query blocks per wave (half the LDS bytes per MFMA, see the kernel); 4 waves = 128 queries per workgroup, 8 waves = 256 for long
    // sequences where the coarser causal diagonal is cheap.  Everything else keeps one block per wave: 8 waves (128 queries) for
    // Lq >= 1024, 4 waves (64 queries) below.  debug word: bit 0 forces <4 waves, 1 block>, bit 1 allows the 8-wave shapes at any
    // length, bit 2 disables the two-block kernels.
    if (Lq <= 8 && S <= 8 && !rel_table && (!(g_attn_dbg & 32) || p.b_inner) && !p.key_valid) {
        const int64_t threads = (int64_t)B * H * 16;
        if (D == 128) attn_tiny_kernel<128><<<(int)((threads + 255) / 256), 256, 0, s>>>(p);
        else attn_tiny_kernel<64><<<(int)((threads + 255) / 256), 256, 0, s>>>(p);
        MC_CHECK_LAUNCH();
        return 0;
    }
#ifdef MC_PROBES
    if (D == 64 && !rel_table && (g_attn_dbg & 16) && Lq > 64) {          // A/B: two query blocks per wave at head_dim 64 (encoders)
        attn_prefill_kernel<64, false, 4, 2><<<dim3((Lq + 127) / 128, H, B), 256, 4 * 64 * 128, s>>>(p);
        MC_CHECK_LAUNCH();
        return 0;
    }
#endif
    const bool two = D == 128 && !rel_table && !(g_attn_dbg & (1 | 4)) && Lq > 64;
    if (two) {
        // 4 waves x 2 query blocks (128 queries per workgroup): after the round-3 clean-up of the softmax's instruction stream the faster shape
        // at every length measured against the 8-wave one (L = 2793, B = 48: 817 vs 800 TFLOP/s; L = 2048: 724 vs 620).  LATE = 1 (round 3):
        // the next tile's LDS-DMA is issued behind K.Q^T, in front of the softmax's VALU-only stretch: +1 %.  (The 8-wave shape, LATE = 0, the
        // bidirectional 32x32x16 build and the round-5 software-pipelined kernel lost their A/Bs - profiles/r03_probes, r05_probes/attn32_ab.json -
        // and are instantiated in the probes build only, or not at all.)
        const dim3 gr((Lq + 127) / 128, H, B);
#ifdef MC_PROBES
        const bool w8 = (g_attn_dbg & 2) && (int64_t)((Lq + 255) / 256) * H * B >= 512 && !(g_attn_dbg & 8);
        if (w8) { attn_prefill_kernel<128, false, 8, 2><<<dim3((Lq + 255) / 256, H, B), 512, 4 * 64 * 256, s>>>(p); MC_CHECK_LAUNCH(); return 0; }
        if (g_attn_dbg & 64) { attn_prefill_kernel<128, false, 4, 2><<<gr, 256, 4 * 64 * 256, s>>>(p); MC_CHECK_LAUNCH(); return 0; }
        if (!p.key_valid && !causal && (g_attn_dbg & 256) && S % 64 == 0) { attn_prefill32_kernel<false><<<gr, 256, 4 * 64 * 256, s>>>(p); MC_CHECK_LAUNCH(); return 0; }
#endif
        if (p.key_valid) attn_prefill_kernel<128, false, 4, 2, false, 1><<<gr, 256, 4 * 64 * 256, s>>>(p);
        else if (S % 64 == 0 && 64LL * k_st * 2 + 256 < (1LL << 31) && 64LL * v_st * 2 + 256 < (1LL << 31)) {   // the LLM's launches: no per-key mask, whole key tiles
            // causal: the 32x32x16 kernel (round 5; measured, profiles/r05_probes/attn32_ab.json: L = 683 +12 %, 2048 +6 %, 2793 +1 %; bidirectional
            // L = 2304 -1.5 %: those keep the 16x16x32 kernel).  Its O stores are 16-byte vectors: rows that are not 16-byte aligned take the
            // 16x16x32 kernel's 8-byte stores.
            const bool o_wide = o_row_stride % 8 == 0 && ((uintptr_t)o & 15) == 0;
            if (causal && o_wide && !(g_attn_dbg & 128)) attn_prefill32_kernel<true><<<gr, 256, 4 * 64 * 256, s>>>(p);
            else attn_prefill_kernel<128, false, 4, 2, false, 1, false, false><<<gr, 256, 4 * 64 * 256, s>>>(p);
        }
        else attn_prefill_kernel<128, false, 4, 2, false, 1, false><<<gr, 256, 4 * 64 * 256, s>>>(p);      // no per-key mask: its loads and branches compiled out
        MC_CHECK_LAUNCH();
        return 0;
    }
    const bool wide = ((g_attn_dbg & 2) || Lq >= 1024) && (int64_t)((Lq + 127) / 128) * H * B >= 512 && !(g_attn_dbg & 1);
    if (wide) {
        dim3 grid((Lq + 127) / 128, H, B);
        if (rel_table) {
            if (D == 128) attn_prefill_kernel<128, true, 8, 1><<<grid, 512, 4 * 64 * 256, s>>>(p);
            else attn_prefill_kernel<64, true, 8, 1><<<grid, 512, 4 * 64 * 128, s>>>(p);
        } else {
            if (D == 128) attn_prefill_kernel<128, false, 8, 1><<<grid, 512, 4 * 64 * 256, s>>>(p);
            else attn_prefill_kernel<64, false, 8, 1><<<grid, 512, 4 * 64 * 128, s>>>(p);
        }
    } else {
        dim3 grid((Lq + 63) / 64, H, B);
        if (rel_table) {
            if (D == 128) attn_prefill_kernel<128, true, 4, 1><<<grid, 256, 4 * 64 * 256, s>>>(p);
            else attn_prefill_kernel<64, true, 4, 1><<<grid, 256, 4 * 64 * 128, s>>>(p);
        } else {
            if (D == 128) attn_prefill_kernel<128, false, 4, 1><<<grid, 256, 4 * 64 * 256, s>>>(p);
            else attn_prefill_kernel<64, false, 4, 1><<<grid, 256, 4 * 64 * 128, s>>>(p);
        }
    }
    MC_CHECK_LAUNCH();
    return 0;
}

// Training forward with dropout on the probabilities (the Q-Former projector of the audio stage-2 recipe: BertSelfAttention.dropout,
// multimodal_projector/Qformer.py:136, :259, attention_probs_dropout_prob = 0.1 of the default BertConfig): the 64-query kernel, lse as in
// mc_attn_prefill_lse_bf16 (of the UN-dropped softmax), masks keyed by (seed, stream_id, element of the [B H Lq, S] score matrix).
extern "C" int mc_attn_prefill_dropout_bf16(const void* q, int64_t q_sb, int64_t q_st, int64_t q_sh, const void* k, int64_t k_sb,
                                            int64_t k_st, int64_t k_sh, const void* v, int64_t v_sb, int64_t v_st, int64_t v_sh,
                                            void* o, int64_t o_row_stride, const int32_t* kv_lens, int B, int H, int Hkv, int Lq, int S, int D,
                                            int causal, int q_offset, float scale, float* lse, float dropout_p, unsigned long long seed,
                                            unsigned int stream_id, const mc_attn_mask* mask, void* stream) {
    const uint8_t* kv_once = nullptr; int64_t kv_once_sb = 0;
    int b_inner_once = 0; int64_t sbi_once = 0;
    mask_fields(mask, kv_once, kv_once_sb, b_inner_once, sbi_once);
    MC_CHECK_ARG(b_inner_once == 0, "mc_attn_prefill_dropout_bf16: a two-level batch index (mc_attn_mask.b_inner) is not defined for this launch");
    MC_CHECK_ARG(q && k && v && o, "mc_attn_prefill_dropout_bf16: null pointer");
    MC_CHECK_ARG(D == 64 || D == 128, "mc_attn_prefill_dropout_bf16: head_dim %d not supported (64 or 128)", D);
    MC_CHECK_ARG(B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && Lq > 0 && S > 0 && S % 4 == 0, "mc_attn_prefill_dropout_bf16: bad shape (S must be a multiple of 4)");
    MC_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "mc_attn_prefill_dropout_bf16: dropout probability has to be between 0 and 1, but got %g", (double)dropout_p);
    MC_CHECK_ARG((q_st % 8 | q_sh % 8 | k_st % 8 | k_sh % 8 | v_st % 8 | v_sh % 8 | q_sb % 8 | k_sb % 8 | v_sb % 8) == 0,
                 "mc_attn_prefill_dropout_bf16: strides must be multiples of 8 elements");
    AttnParams p{(const bf16_t*)q, q_sb, q_st, q_sh, (const bf16_t*)k, k_sb, k_st, k_sh, (const bf16_t*)v, v_sb, v_st, v_sh,
                 (bf16_t*)o, o_row_stride, nullptr, kv_lens, B, H, Hkv, Lq, S, causal, q_offset,
                 scale * 1.4426950408889634f, nullptr, nullptr, 0, 0, lse};
    const double t = (double)dropout_p * 4294967296.0;
    p.drop = AttnDropout{t >= 4294967295.0 ? 4294967295u : (uint32_t)t, (uint32_t)seed, (uint32_t)(seed >> 32), stream_id, 1.0f / (1.0f - dropout_p)};
    p.key_valid = kv_once; p.key_valid_sb = kv_once_sb;
    dim3 grid((Lq + 63) / 64, H, B);
    hipStream_t s = (hipStream_t)stream;
    if (D == 128) attn_prefill_kernel<128, false, 4, 1, true><<<grid, 256, 4 * 64 * 256, s>>>(p);
    else attn_prefill_kernel<64, false, 4, 1, true><<<grid, 256, 4 * 64 * 128, s>>>(p);
    MC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mc_attn_prefill_bf16(const void* q, int64_t q_sb, int64_t q_st, int64_t q_sh, const void* k, int64_t k_sb,
                                    int64_t k_st, int64_t k_sh, const void* v, int64_t v_sb, int64_t v_st, int64_t v_sh,
                                    void* o, int64_t o_row_stride, const int32_t* out_map, const int32_t* kv_lens, int B,
                                    int H, int Hkv, int Lq, int S, int D, int causal, int q_offset, float scale,
                                    const float* rel_table, int rel_stride, int rel_off, const float* q_gate, const mc_attn_mask* mask,
                                    void* stream) {
    return mc_attn_prefill_lse_bf16(q, q_sb, q_st, q_sh, k, k_sb, k_st, k_sh, v, v_sb, v_st, v_sh, o, o_row_stride, out_map, kv_lens, B,
                                    H, Hkv, Lq, S, D, causal, q_offset, scale, rel_table, rel_stride, rel_off, q_gate, nullptr, mask, stream);
}

// ------------------------------------------------------------------------------------------
// output_attentions: the probabilities themselves (`attn_weights` of LocalLoraAttention.forward, multimodal_llama.py:295-312).  Not on the
// generation path - the flash kernels above never materialise them - so this is the plain form: one wave per (b, h, query) row, a key per
// lane, two sweeps over the keys (online max / sum, then exp(s - max) / sum rounded to the storage type).  Masked entries are exact zeros.
struct ProbsParams {
    const bf16_t* q; int64_t q_sb, q_st, q_sh; const bf16_t* k; int64_t k_sb, k_st, k_sh;
    const int32_t* kv_lens; const uint8_t* key_valid; int64_t kv_stride; bf16_t* out;
    int B, H, Hkv, Lq, S, causal, q_offset; float scale_log2;
};
template <int D>
__global__ __launch_bounds__(256) void attn_probs_kernel(ProbsParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t qs[4][D];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4LL + wave;
    const bool on = row < (int64_t)p.B * p.H * p.Lq;
    const int64_t rr = on ? row : 0;
    const int i = (int)(rr % p.Lq), h = (int)((rr / p.Lq) % p.H), b = (int)(rr / p.Lq / p.H);
    const int hk = h / (p.H / p.Hkv);
    if (lane < D / 8) *(bf16x8*)&qs[wave][lane * 8] = *(const bf16x8*)(p.q + b * p.q_sb + (int64_t)i * p.q_st + h * p.q_sh + lane * 8);
    __syncthreads();
    if (!on) return;
    int limit = p.kv_lens ? min(p.kv_lens[b], p.S) : p.S;
    if (p.causal) limit = min(limit, i + p.q_offset + 1);
    const bf16_t* kb = p.k + b * p.k_sb + hk * p.k_sh;
    const uint8_t* kv = p.key_valid ? p.key_valid + b * p.kv_stride : nullptr;
    auto score = [&](int j) -> float {
        const bf16_t* kr = kb + (int64_t)j * p.k_st;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < D / 8; ++c) {
            const bf16x8 kk = *(const bf16x8*)(kr + c * 8), qq = *(const bf16x8*)&qs[wave][c * 8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf((float)qq[e], (float)kk[e], a);
        }
        return a * p.scale_log2;
    };
    float m = NEG_BIG, l = 0.f;
    for (int j = lane; j < limit; j += 64) {
        if (kv && !kv[j]) continue;
        const float sc = score(j);
        const float mn = fmaxf(m, sc);
        l = l * fast_exp2(m - mn) + fast_exp2(sc - mn);
        m = mn;
    }
    const float mw = wave_max(m);
    l = wave_sum(l * (m > NEG_BIG ? fast_exp2(m - mw) : 0.f));
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    bf16_t* orow = p.out + row * (int64_t)p.S;
    for (int j = lane; j < p.S; j += 64) {
        float pr = 0.f;
        if (j < limit && !(kv && !kv[j])) pr = fast_exp2(score(j) - mw) * inv;
        orow[j] = (bf16_t)pr;
    }
}

extern "C" int mc_attn_probs_bf16(const void* q, int64_t q_sb, int64_t q_st, int64_t q_sh, const void* k, int64_t k_sb, int64_t k_st, int64_t k_sh,
                                  const int32_t* kv_lens, const void* key_valid, int64_t kv_stride, void* probs, int B, int H, int Hkv, int Lq, int S,
                                  int D, int causal, int q_offset, float scale, void* stream) {
    MC_CHECK_ARG(q && k && probs, "mc_attn_probs_bf16: null pointer");
    MC_CHECK_ARG(D == 64 || D == 128, "mc_attn_probs_bf16: head_dim %d not supported (64 or 128)", D);
    MC_CHECK_ARG(B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && Lq > 0 && S > 0, "mc_attn_probs_bf16: bad shape");
    MC_CHECK_ARG((q_st % 8 | q_sh % 8 | k_st % 8 | k_sh % 8 | q_sb % 8 | k_sb % 8) == 0, "mc_attn_probs_bf16: strides must be multiples of 8 elements");
    MC_CHECK_ARG(!key_valid || kv_stride >= S, "mc_attn_probs_bf16: key mask rows shorter than S");
    ProbsParams p{(const bf16_t*)q, q_sb, q_st, q_sh, (const bf16_t*)k, k_sb, k_st, k_sh, kv_lens, (const uint8_t*)key_valid, kv_stride,
                  (bf16_t*)probs, B, H, Hkv, Lq, S, causal, q_offset, scale * 1.4426950408889634f};
    const int64_t rows = (int64_t)B * H * Lq;
    MC_CHECK_ARG((rows + 3) / 4 < (1LL << 31), "mc_attn_probs_bf16: too many rows");
    if (D == 128) attn_probs_kernel<128><<<(int)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(p);
    else attn_probs_kernel<64><<<(int)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(p);
    MC_CHECK_LAUNCH();
    return 0;
}

// workspace of the decode kernels: one partial (m, l, acc[D]) per (b, h) and chunk of kDecChunk keys of a cache of S positions
extern "C" int mc_attn_decode_workspace_bytes(int B, int H, int D, int S, int64_t* bytes) {
    *bytes = (int64_t)B * H * ((S + kDecChunk - 1) / kDecChunk + 1) * (D + 2) * 4;
    return 0;
}

// nsplit workgroups per head (group) share its chunks (a performance knob: the result does not depend on it).  The workgroup combines the
// partials itself when it owns all of them and they fit its LDS; otherwise they go through `workspace` (mc_attn_decode_workspace_bytes).
static int decode_launch(DecodeParams& p, int D, int nsplit, hipStream_t s, const char* who) {
    const int chunks = (p.S + kDecChunk - 1) / kDecChunk + 1;       // + the fused token's own partial
    nsplit = max(1, min(nsplit, chunks));                           // more than one workgroup per chunk has nothing to do
    p.ws_chunks = chunks;
    p.ws_lds = (nsplit == 1 && chunks <= kDecLdsChunks) ? 1 : 0;
    MC_CHECK_ARG(p.ws_lds || p.ws, "%s: more than one workgroup per head, or a cache of more than %d positions, needs a workspace", who,
                 (kDecLdsChunks - 1) * kDecChunk);
    const size_t lds = p.ws_lds ? (size_t)chunks * (D + 2) * sizeof(float) : 0;
    const dim3 grid(p.B * p.H, nsplit);
    const bool deep = (int64_t)grid.x * grid.y < 1024;               // fewer than one workgroup per SIMD: keep more bytes in flight per wave
    if (D == 128) {
        if (deep) attn_decode_kernel<128, 8><<<grid, 256, lds, s>>>(p);
        else attn_decode_kernel<128, 4><<<grid, 256, lds, s>>>(p);
        if (!p.ws_lds) attn_decode_combine_kernel<128><<<p.B * p.H, 128, 0, s>>>(p);
    } else {
        if (deep) attn_decode_kernel<64, 8><<<grid, 256, lds, s>>>(p);
        else attn_decode_kernel<64, 4><<<grid, 256, lds, s>>>(p);
        if (!p.ws_lds) attn_decode_combine_kernel<64><<<p.B * p.H, 64, 0, s>>>(p);
    }
    MC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mc_attn_decode_bf16(const void* q, int64_t q_sb, int64_t q_sh, const void* k, int64_t k_sb, int64_t k_st,
                                   int64_t k_sh, const void* v, int64_t v_sb, int64_t v_st, int64_t v_sh, void* o,
                                   int64_t o_sb, void* workspace, const int32_t* kv_lens, int B, int H, int Hkv, int S,
                                   int D, int nsplit, float scale, const mc_attn_mask* mask, void* stream) {
    MC_CHECK_ARG(q && k && v && o, "mc_attn_decode_bf16: null pointer");
    MC_CHECK_ARG(D == 64 || D == 128, "mc_attn_decode_bf16: head_dim %d not supported (64 or 128)", D);
    MC_CHECK_ARG(nsplit >= 1, "mc_attn_decode_bf16: nsplit must be >= 1");
    MC_CHECK_ARG(B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && S > 0, "mc_attn_decode_bf16: bad shape");
    MC_CHECK_ARG(!mask || mask->b_inner == 0, "mc_attn_decode_bf16: no two-level batch index here");
    DecodeParams p{(const bf16_t*)q, q_sb, q_sh, (const bf16_t*)k, k_sb, k_st, k_sh, (const bf16_t*)v, v_sb, v_st, v_sh,
                   (bf16_t*)o, o_sb, (float*)workspace, kv_lens, B, H, Hkv, S, nsplit, scale * 1.4426950408889634f,
                   nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    p.key_valid = mask ? (const uint8_t*)mask->key_valid : nullptr; p.key_valid_sb = mask && mask->key_valid ? mask->key_valid_stride : 0;
    return decode_launch(p, D, nsplit, (hipStream_t)stream, "mc_attn_decode_bf16");
}

// Decode attention with RoPE and the KV-cache append fused in: qkv [B, (H + 2 Hkv) * D] is the pre-rotary output of the q|k|v linear
// for the token at position kv_lens[b] - 1 of every sequence (kv_lens counts this token).  Replaces mc_rope_kv_bf16 +
// mc_attn_decode_bf16 for one-token steps (multimodal_llama.py:281-312): one launch less per layer and no q round trip.
extern "C" int mc_attn_decode_rope_bf16(const void* qkv, int64_t qkv_ld, const float* cos_table, const float* sin_table, void* k_cache,
                                        int64_t k_sb, int64_t k_st, int64_t k_sh, void* v_cache, int64_t v_sb, int64_t v_st, int64_t v_sh,
                                        void* o, int64_t o_sb, void* workspace, const int32_t* kv_lens, int B, int H, int Hkv, int S, int D,
                                        int nsplit, float scale, const mc_attn_mask* mask, void* stream) {
    MC_CHECK_ARG(qkv && cos_table && sin_table && k_cache && v_cache && o && kv_lens, "mc_attn_decode_rope_bf16: null pointer");
    MC_CHECK_ARG(D == 64 || D == 128, "mc_attn_decode_rope_bf16: head_dim %d not supported (64 or 128)", D);
    MC_CHECK_ARG(nsplit >= 1, "mc_attn_decode_rope_bf16: nsplit must be >= 1");
    MC_CHECK_ARG(B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && S > 0 && qkv_ld % 8 == 0, "mc_attn_decode_rope_bf16: bad shape");
    MC_CHECK_ARG(!mask || mask->b_inner == 0, "mc_attn_decode_rope_bf16: no two-level batch index here");
    DecodeParams p{nullptr, 0, 0, (const bf16_t*)k_cache, k_sb, k_st, k_sh, (const bf16_t*)v_cache, v_sb, v_st, v_sh,
                   (bf16_t*)o, o_sb, (float*)workspace, kv_lens, B, H, Hkv, S, nsplit, scale * 1.4426950408889634f,
                   (const bf16_t*)qkv, qkv_ld, cos_table, sin_table, (bf16_t*)k_cache, (bf16_t*)v_cache};
    p.key_valid = mask ? (const uint8_t*)mask->key_valid : nullptr; p.key_valid_sb = mask && mask->key_valid ? mask->key_valid_stride : 0;
    return decode_launch(p, D, nsplit, (hipStream_t)stream, "mc_attn_decode_rope_bf16");
}
