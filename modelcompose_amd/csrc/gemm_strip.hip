// Strip kernel: every linear of at most 64 rows (the decode step's projections, the last-token tail of a prefill, lm_head).  HBM-bound.
//
// Replaces, for M <= 64, the F.linear calls of modelcompose/model/language_model/multimodal_llama.py:122 (LocalLoraLinear base GEMM with
// the default adapter only during decode, :435-438), :262-268 / :335-336 / :380-394 (q|k|v, o, gate|up, down) and :720 (lm_head).
//
// One K-reduction order per (N, K), whatever M is.  A workgroup owns a strip of R weight block-rows (16 R output columns) over the WHOLE
// K range; its 8 waves split K into 8 contiguous chunks (strip_chunk() below: a function of K only), every wave runs ONE chain of
// v_mfma_f32_16x16x32 over its chunk in ascending k from a zero accumulator, and the 8 chains are added in a fixed tree
// ((w0 + w4) + (w1 + w5)) + (w2 + w6)) + (w3 + w7).  Neither R (picked from N and the number of row blocks), nor the number of row blocks
// MB = ceil(M / 16), nor the prefetch depth U changes which products enter which chain or the order of the additions, and a row of an
// MFMA result depends on that row of x only: row m of the output is the same bits at M = 1, 8, 47, 48 or 64 - the property the eval
// loader's per-rank batching needs (model_multimodal_qa_loader.py:25-46: a question sits in a different batch at 1 and at 8 GPUs).
// There is no split of K over workgroups, hence no fp32 slabs, no reduce launch and no workspace.
//
// Data movement: weights go HBM -> VGPR once (nontemporal 1-KiB fragment loads, packed layout of gemm.hip), x L2 -> VGPR once per workgroup
// in whole lines (kernel comment); both sit in a ring of U k-steps per wave that is refilled slot by slot right after the slot's MFMAs (4 U (R + MB)
// registers; U (R + MB) KiB in flight per wave, 8 waves per CU).  The chunk lengths are multiples of U by construction (host check), so
// the loop has no conditional load and the compiler's vmcnt accounting is exact.  RMSNorm's row factor (rms_eps) is computed from the x
// fragments as they pass: per lane in k order, then over the 4 k-quarters of a row, then over the 8 waves in wave order.
#include "gemm_epilogue.h"

namespace {

constexpr int kStripWaves = 8;

// chunk of wave w: k-blocks [start, start + len) of the kblocks 32-column blocks of K.  Units of G = 4 k-blocks when kblocks allows
// (K % 128 == 0), else 2 (K is a multiple of 64); the first (units % 8) waves take one unit more.  A function of K only.
__host__ __device__ inline void strip_chunk(int kblocks, int w, int& start, int& len) {
    const int G = (kblocks & 3) == 0 ? 4 : 2;
    const int units = kblocks / G, q = units / kStripWaves, rm = units % kStripWaves;
    start = G * (w * q + (w < rm ? w : rm));
    len = G * (q + (w < rm ? 1 : 0));
}

// x fragments come through a wave-private LDS scratch: a load instruction reads WHOLE 128-byte lines (8 rows x 128 bytes: lane l takes
// 16-byte piece l & 7 of row l >> 3 of its half block) - the fragment order of the MFMA operand (16 rows x 64 bytes per instruction) uses
// half of every line it touches, and with ~100 KiB of x in flight per CU the other half has left the 32 KiB L1 before the next k-step
// asks for it: measured 31 GB/s per CU of intake whatever the source, i.e. x (3x the weight bytes at M = 48, N = 4096) set the time.  The
// pieces of a k-step PAIR (64 columns = one line per row) are written to LDS as they sit in the registers and read back in fragment
// order; unit (row8, piece) of a half block lives at 16-byte slot row8 * 8 + (piece ^ 2 (row8 >> 1)), which makes both the 8-lane
// groups of the ds_write_b128 and the 16-lane groups of the ds_read_b128 conflict-free.  A wave's LDS instructions execute in order, the
// scratch is its own: no barrier, no wait beyond the read's own.
template <int MB, int R, int U>
__global__ __launch_bounds__(kStripWaves * 64) void gemm_strip_kernel(const bf16_t* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ wp,
                                                                      int M, int N, int K, Epilogue ep) {
    static_assert(U % 2 == 0, "the x ring holds k-step pairs");
    constexpr int UP = U / 2;
    constexpr int kRedFloats = 4 * R * MB * 256, kScrFloats = kStripWaves * MB * 2 * 256;
    __shared__ __attribute__((aligned(16))) float smem[kRedFloats > kScrFloats ? kRedFloats : kScrFloats];
    __shared__ float redss[kStripWaves][MB][16];
    float (*red)[R * MB][64][4] = (float (*)[R * MB][64][4])smem;          // after the main loop (behind a barrier)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;
    const int kblocks = K >> 5, nblocks = (N + 15) >> 4;
    const int nb0 = blockIdx.x * R;
    int kb0, len;
    strip_chunk(kblocks, wave, kb0, len);

    const bf16_t* wptr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wptr[r] = wp + ((int64_t)min(nb0 + r, nblocks - 1) * kblocks + kb0) * 512 + lane * 8;
    // line loads: row8 = lane >> 3, piece = lane & 7 of half block h of row block b
    const int row8 = lane >> 3, piece = lane & 7;
    const bf16_t* xptr[MB][2];
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h) xptr[b][h] = x + (int64_t)min(b * 16 + h * 8 + row8, M - 1) * ldx + kb0 * 32 + piece * 8;
    bf16x8* scr = (bf16x8*)smem + wave * (MB * 2 * 64);                     // this wave's scratch: [MB][2][64] 16-byte slots
    const int wslot = row8 * 8 + (piece ^ (2 * (row8 >> 1)));
    int rslot[2];                                                           // fragment read of k-step s of the pair: row c16, piece 4 s + q4
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) rslot[s_] = (c16 >> 3) * 64 + (c16 & 7) * 8 + ((4 * s_ + q4) ^ (2 * ((c16 & 7) >> 1)));

    f32x4 acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[r][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float ss[MB];                                  // this lane's share of sum_k x[m][k]^2 (row m = 16 b + c16, its 8-element k pieces)
#pragma unroll
    for (int b = 0; b < MB; ++b) ss[b] = 0.f;

    bf16x8 wf[U][R], xr[UP][MB][2];
    auto load_w = [&](int s_, int k) {
#pragma unroll
        for (int r = 0; r < R; ++r) wf[s_][r] = __builtin_nontemporal_load((const bf16x8*)(wptr[r] + (int64_t)k * 512));
    };
    auto load_x = [&](int p_, int k) {             // the pair of k-steps k, k + 1
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) xr[p_][b][h] = *(const bf16x8*)(xptr[b][h] + k * 32);
    };
    auto use_pair = [&](int p_) {
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) scr[(b * 2 + h) * 64 + wslot] = xr[p_][b][h];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            bf16x8 xf[MB];
#pragma unroll
            for (int b = 0; b < MB; ++b) xf[b] = scr[b * 128 + rslot[s_]];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int b = 0; b < MB; ++b) acc[r][b] = mc_mfma_16x16x32(wf[2 * p_ + s_][r], xf[b], acc[r][b]);
            // (unconditional: a branch in the streaming loop costs the load pipelining far more than these FMAs)
#pragma unroll
            for (int b = 0; b < MB; ++b)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = (float)xf[b][j]; ss[b] = fmaf(f, f, ss[b]); }
        }
    };
    if (len > 0) {                                 // len is a multiple of U (host check)
#pragma unroll
        for (int j = 0; j < UP; ++j) { load_w(2 * j, 2 * j); load_w(2 * j + 1, 2 * j + 1); load_x(j, 2 * j); }
        int k = 0;
        for (; k + U < len; k += U) {
#pragma unroll
            for (int j = 0; j < UP; ++j) {
                use_pair(j);
                load_w(2 * j, k + U + 2 * j); load_w(2 * j + 1, k + U + 2 * j + 1); load_x(j, k + U + 2 * j);
            }
        }
#pragma unroll
        for (int j = 0; j < UP; ++j) use_pair(j);
    }
    __syncthreads();                               // every wave is done with its scratch: the buffer becomes the reduction's

    // ---- the 8 chains: (w + (w + 4)) for w = 0..3, then those four in order
    if (wave >= 4) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < MB; ++b) *(f32x4*)&red[wave - 4][r * MB + b][lane][0] = acc[r][b];
    }
    const bool want_rms = ep.rms_eps > 0.f;
    if (want_rms) {
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            float v = ss[b];
            v += __shfl_xor(v, 16, 64);                       // the four k pieces (q4) of the row
            v += __shfl_xor(v, 32, 64);
            if (q4 == 0) redss[wave][b][c16] = v;
        }
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < MB; ++b) {         // (slot `wave` is read and rewritten by this wave only)
                acc[r][b] += *(f32x4*)&red[wave][r * MB + b][lane][0];
                *(f32x4*)&red[wave][r * MB + b][lane][0] = acc[r][b];
            }
    }
    __syncthreads();
    // row factor of row m = 16 b + c16 (the row every store below gives this lane): waves summed in wave order
    auto row_factor = [&](int b) -> float {
        if (!want_rms) return 1.0f;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < kStripWaves; ++w) t += redss[w][b][c16];
        return rsqrtf(t / (float)K + ep.rms_eps);
    };
    auto total = [&](int p) -> f32x4 {
        f32x4 s_ = *(f32x4*)&red[0][p][lane][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) s_ += *(f32x4*)&red[w][p][lane][0];
        return s_;
    };
    if (ep.swiglu) {
        if constexpr (R % 2 == 0) {
            for (int p = wave; p < (R / 2) * MB; p += kStripWaves) {
                const int rp = p / MB, b = p - rp * MB;
                const f32x4 g = total((2 * rp) * MB + b), u = total((2 * rp + 1) * MB + b);
                const int m = b * 16 + c16;
                const int nb = nb0 + 2 * rp;
                Epilogue e2 = ep;
                e2.alpha = ep.alpha * row_factor(b);
                if (m < M && (nb + 1) * 16 < N) epilogue_store4_swiglu(e2, m, (nb >> 1) * 16 + q4 * 4, g, u);
            }
        }
        return;
    }
    for (int p = wave; p < R * MB; p += kStripWaves) {
        const int r = p / MB, b = p - r * MB;
        const f32x4 s_ = total(p);
        const int m = b * 16 + c16;
        const int n = (nb0 + r) * 16 + q4 * 4;
        Epilogue e2 = ep;
        e2.alpha = ep.alpha * row_factor(b);
        if (m < M && n < N) epilogue_store4(e2, m, n, s_);
    }
}

// block-rows per workgroup: one workgroup per CU and round, a round costs the bytes its workgroup takes in (R weight block-rows from HBM,
// MB activation block-rows from L2, counted at half weight); the fewest round-bytes wins, ties go to the larger R.  R * MB <= 24 keeps the
// accumulators + rings inside 256 registers and the reduction buffer (4 R MB KiB) inside a third of the LDS.  Results do not depend on R.
int strip_rows(int nblocks, int mb, bool swiglu) {
    static const int cand[6] = {8, 6, 4, 3, 2, 1};
    int best = swiglu ? 2 : 1;
    double best_cost = 1e30;
    for (int i = 0; i < 6; ++i) {
        const int R = cand[i];
        if (R * mb > 24 || (R == 8 && mb > 2)) continue;
        if (swiglu && (R & 1)) continue;
        const int64_t wgs = (nblocks + R - 1) / R;
        const int64_t rounds = (wgs + 255) / 256;
        const double cost = (double)rounds * (R + 0.5 * mb);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = R; }
    }
    return best;
}

template <int MB, int R, int U>
void launch_u(int grid, hipStream_t s, const bf16_t* x, int64_t ldx, const bf16_t* w, int M, int N, int K, const Epilogue& ep) {
    gemm_strip_kernel<MB, R, U><<<grid, kStripWaves * 64, 0, s>>>(x, ldx, w, M, N, K, ep);
}

// ring depth: the deepest of {8, 4, 2} that divides every wave's chunk and keeps 4 U (R + MB) ring registers + 4 R MB accumulators <= 144 (the
// compiler's schedule adds up to ~100 registers of its own on top: builds above that spilled)
template <int MB, int R>
void launch_r(int umax, int grid, hipStream_t s, const bf16_t* x, int64_t ldx, const bf16_t* w, int M, int N, int K, const Epilogue& ep) {
    constexpr int budget = 144 - 4 * R * MB;
    if constexpr (4 * 8 * (R + MB) <= budget) { if (umax >= 8) { launch_u<MB, R, 8>(grid, s, x, ldx, w, M, N, K, ep); return; } }
    if constexpr (4 * 4 * (R + MB) <= budget) { if (umax >= 4) { launch_u<MB, R, 4>(grid, s, x, ldx, w, M, N, K, ep); return; } }
    launch_u<MB, R, 2>(grid, s, x, ldx, w, M, N, K, ep);
}

template <int MB>
void launch_mb(int R, int umax, int grid, hipStream_t s, const bf16_t* x, int64_t ldx, const bf16_t* w, int M, int N, int K, const Epilogue& ep) {
    switch (R) {
        case 1: launch_r<MB, 1>(umax, grid, s, x, ldx, w, M, N, K, ep); break;
        case 2: launch_r<MB, 2>(umax, grid, s, x, ldx, w, M, N, K, ep); break;
        case 3: launch_r<MB, 3>(umax, grid, s, x, ldx, w, M, N, K, ep); break;
        case 4: launch_r<MB, 4>(umax, grid, s, x, ldx, w, M, N, K, ep); break;
        case 6: launch_r<MB, 6>(umax, grid, s, x, ldx, w, M, N, K, ep); break;
        default:
            if constexpr (8 * MB <= 16) launch_r<MB, 8>(umax, grid, s, x, ldx, w, M, N, K, ep);
            break;
    }
}

}  // namespace

int mc_strip_launch(const bf16_t* x, int64_t ldx, const bf16_t* w_packed, int M, int N, int K, const Epilogue& ep, hipStream_t s) {
    // (the dispatcher in gemm.hip slices longer launches; K is the packed K, a multiple of 64)
    MC_CHECK_ARG(M >= 1 && M <= 64 && N >= 1 && K >= 64 && K % 64 == 0, "strip GEMM: M %d (1..64), N %d, K %d (a multiple of 64) out of range", M, N, K);
    const int mb = (M + 15) / 16, nblocks = (N + 15) / 16, kblocks = K >> 5;
    const int R = strip_rows(nblocks, mb, ep.swiglu != 0);
    // the deepest ring every wave's chunk length is a multiple of
    int umax = 8;
    for (int w = 0; w < kStripWaves; ++w) {
        int st, len;
        strip_chunk(kblocks, w, st, len);
        while (umax > 2 && len % umax) umax >>= 1;
    }
    const int grid = (nblocks + R - 1) / R;
    switch (mb) {
        case 1: launch_mb<1>(R, umax, grid, s, x, ldx, w_packed, M, N, K, ep); break;
        case 2: launch_mb<2>(R, umax, grid, s, x, ldx, w_packed, M, N, K, ep); break;
        case 3: launch_mb<3>(R, umax, grid, s, x, ldx, w_packed, M, N, K, ep); break;
        default: launch_mb<4>(R, umax, grid, s, x, ldx, w_packed, M, N, K, ep); break;
    }
    return 0;
}
