// bf16 linear layers for gfx950:  out[M,N] = act(x[M,K] · W[N,K]^T + bias) (+ residual)
//
// Replaces every F.linear on the hot path of the reference
// (modelcompose/model/language_model/multimodal_llama.py:122 LocalLoraLinear base GEMM,
//  :262-268/:335-336/:380-394 projections, :720 lm_head; CLIP / projector linears).
//
// HBM layout of a weight ("packed"): the row-major [N,K] checkpoint tensor is re-tiled once at
// load time into 1-KiB MFMA fragments:  block (nb, kb) covers rows 16nb..16nb+15, cols 32kb..32kb+31
// and stores lane l = (q<<4 | r) -> 8 contiguous bf16 = W[16nb + r][32kb + 8q .. +7].
// Blocks are ordered nb-major, kb-minor.  One 16-byte-per-lane wave load (1 KiB, fully coalesced)
// is then exactly the A operand of v_mfma_f32_16x16x32_bf16, for the kernels of both families (mc_gemm_args.family, include/mc_hip.h):
//   TILE  * gemm_tile_kernel    (MFMA-bound): 128x128x64 tiles, LDS-DMA (global_load_lds) double buffer.
//         * gemm_tile256_kernel (enough 256x256 tiles to fill the chip): two wave groups alternating LDS loads with MFMA work;
//           <., 3>: 192-column tiles for under-filled launches.  The three give bit-identical outputs (one K order).
//   STRIP * gemm_strip_kernel (gemm_strip.hip; launches of at most 64 rows, HBM-bound): weights streamed straight to VGPRs, K split over
//           the 8 waves of a workgroup, one K order per (N, K) whatever M is.
// MFMA roles: A = weight fragment (rows = n), B = activation fragment (cols = m) so that a lane
// ends up with 4 consecutive n for one token m  ->  8-byte packed bf16 stores.
#include "common.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int N, int K, int Np,
                                   int Kp, int64_t ldw, int64_t sk = 1) {
    // one thread per 16-byte fragment piece
    const int64_t nfrag = (int64_t)(Np / 16) * (Kp / 32) * 64;
    for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < nfrag; f += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(f & 63);
        const int64_t blk = f >> 6;
        const int kb = (int)(blk % (Kp / 32));
        const int nb = (int)(blk / (Kp / 32));
        const int n = nb * 16 + (lane & 15);
        const int k = kb * 32 + (lane >> 4) * 8;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (n < N && k + j < K) ? w[(int64_t)n * ldw + (k + j) * sk] : (bf16_t)0.0f;
        *(bf16x8*)(out + f * 8) = v;
    }
}

// Many packs in one launch (the finetune step refreshes ~1400 LoRA fragment images after every optimizer step): blockIdx.y = descriptor.
struct PackDesc { const bf16_t* src; bf16_t* dst; int64_t sn, sk; int N, K; };
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const PackDesc* __restrict__ descs) {
    const PackDesc d = descs[blockIdx.y];
    const int Np = (d.N + 15) / 16 * 16, Kp = (d.K + 63) / 64 * 64;
    const int kbn = Kp / 32;
    const int64_t nfrag = (int64_t)(Np / 16) * kbn * 64;
    for (int64_t f = blockIdx.x * 256LL + threadIdx.x; f < nfrag; f += (int64_t)gridDim.x * 256) {
        const int lane = (int)(f & 63);
        const int64_t blk = f >> 6;
        const int kb = (int)(blk % kbn), nb = (int)(blk / kbn);
        const int n = nb * 16 + (lane & 15);
        const int k = kb * 32 + (lane >> 4) * 8;
        bf16x8 v;
        if (d.sk == 1 && n < d.N && k + 8 <= d.K && ((d.sn & 7) == 0)) {
            v = *(const bf16x8*)(d.src + (int64_t)n * d.sn + k);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (n < d.N && k + j < d.K) ? d.src[(int64_t)n * d.sn + (k + j) * d.sk] : (bf16_t)0.0f;
        }
        *(bf16x8*)(d.dst + f * 8) = v;
    }
}

// descs: n descriptors in DEVICE memory {src, dst, stride_n, stride_k, N, K} (layout of struct PackDesc: 2 pointers, 2 int64, 2 int32)
extern "C" int mc_pack_weight_batch_bf16(const void* descs_dev, int n, int blocks_per_desc, void* stream) {
    MC_CHECK_ARG(descs_dev && n > 0 && blocks_per_desc > 0, "mc_pack_weight_batch_bf16: bad arguments");
    for (int i = 0; i < n; i += 65535) {
        const int cnt = min(65535, n - i);
        pack_weight_batch_kernel<<<dim3(blocks_per_desc, cnt), 256, 0, (hipStream_t)stream>>>((const PackDesc*)descs_dev + i);
    }
    MC_CHECK_LAUNCH();
    return 0;
}

__global__ void unpack_weight_kernel(const bf16_t* __restrict__ p, bf16_t* __restrict__ w, int N, int K, int Kp) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / K), k = (int)(i % K);
        const int64_t blk = (int64_t)(n >> 4) * (Kp / 32) + (k >> 5);
        const int lane = (((k & 31) >> 3) << 4) | (n & 15);
        w[i] = p[blk * 512 + lane * 8 + (k & 7)];
    }
}

#include "gemm_epilogue.h"

// ---- wide epilogue of the 256x256 kernel ------------------------------------------------------------------------------------------
// A lane of the 16x16x32 accumulator layout owns 4 consecutive columns of a 16-column block (lane = q4*16 + c16: row c16, columns
// 4*q4 .. 4*q4+3), i.e. 8 bytes of bf16 - an 8-byte store per block, 32 per wave.  v_permlane16_swap exchanges the packed halves of
// two neighbouring blocks between the lane pairs (q4 = 0,1) and (q4 = 2,3) of a row, after which every lane holds 8 consecutive
// columns of ONE block: a 16-byte store per block PAIR (half the store instructions, same bytes; the tail of a tile is store-issue
// bound, cdna_hip_programming.md T21).  The values are computed exactly as epilogue_store4 / _swiglu compute them.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the four epilogue values of one accumulator fragment; a = alpha * row_scale[m] (hoisted by the caller), res = the residual's 4 values
// ACTC: how the activation is evaluated - 0 none, 1 the sigmoid family x sigmoid(k x) (QuickGELU k = 1.702, SiLU k = 1), 3 exact GELU, 2 the
// generic per-element switch.  The caller picks the class once per row: with the switch inside the element loops every element walked a scalar
// branch tree (a CLIP fc1 launch, K = 1024: +34 % over the same launch without activation; +13 % with the class hoisted).
// BIASC / RESC: 1 / 0 = known present / absent (the branch is hoisted by the caller), -1 = decided here per call.
template <int ACTC = 2, int BIASC = -1, int RESC = -1>
__device__ __forceinline__ bf16x4 epilogue_vals4(const Epilogue& e, float a, int n, f32x4 v, bool has_res, bf16x4 res, float act_k = 1.0f) {
    float r[4] = {v[0] * a, v[1] * a, v[2] * a, v[3] * a};
    if (BIASC == 1 || (BIASC == -1 && e.bias)) {
        bf16x4 b = *(const bf16x4*)(e.bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] += (float)b[i];
    }
    if constexpr (ACTC == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = r[i] * mc_sigmoid(act_k * r[i]);
    } else if constexpr (ACTC == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = mc_gelu(r[i]);
    } else if constexpr (ACTC == 2) {
        if (e.act != MC_ACT_NONE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = mc_act(r[i], e.act);
        }
    }
    if (RESC == 1 || (RESC == -1 && has_res)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] += e.beta * (float)res[i];
    }
    return (bf16x4){(bf16_t)r[0], (bf16_t)r[1], (bf16_t)r[2], (bf16_t)r[3]};
}

__device__ __forceinline__ bf16x4 swiglu_vals4(float a, f32x4 g, f32x4 u) {
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float gg = (float)(bf16_t)(g[i] * a), uu = (float)(bf16_t)(u[i] * a);
        o[i] = (bf16_t)(gg * mc_sigmoid(gg) * uu);
    }
    return o;
}

// lo = this lane's 4 columns of block b, hi = of block b+1 (both for row m).  Stores 16 bytes: lanes with even q4 the columns
// 8*(q4>>1) .. +7 of block b, odd q4 the same columns of block b+1.  row = &out[m][first column of block b]; blocks are 16 columns apart.
__device__ __forceinline__ void store_pair16(bf16_t* row, int q4, bf16x4 lo, bf16x4 hi) {
    u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi);
    auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
    *(u32x4*)(row + (q4 & 1) * 16 + (q4 >> 1) * 8) = o;
}

// ------------------------------------------------------------------------------------------
// large-M kernel
// ------------------------------------------------------------------------------------------
#define TN 128   // weight rows per tile
#define TM 128   // tokens per tile
#define TK 64
#define TILE_BYTES (TN * TK * 2)   // 16 KiB per operand per stage

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__global__ __launch_bounds__(256, 2) void gemm_tile_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                           const bf16_t* __restrict__ wp, int M, int N, int K,
                                                           Epilogue ep, int tiles_m, int tiles_n, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [stage][W 16K | X 16K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = wave >> 1, wave_m = wave & 1;

    // --- XCD-aware tile mapping: blocks b and b+8 share an XCD (round-robin dispatch), give each XCD a
    // contiguous chunk of the logical tile order, then walk tiles in groups of 8 m-tiles per n-tile column.
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int g = bid / per_group;
    const int first_m = g * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int tm = first_m + (bid % per_group) % gsz;
    const int tn = (bid % per_group) / gsz;
    const int m0 = (dbg & 1) ? 0 : tm * TM, n0 = (dbg & 1) ? 0 : tn * TN;

    const int kblocks = K >> 5;           // 32-wide k blocks in the packed weight
    const int nblocks = (N + 15) >> 4;
    // split-K (gridDim.y > 1): this workgroup reduces K-tiles [t_begin, t_end) and stores raw fp32 partial sums into slab blockIdx.y
    const int nt_all = K / TK;
    const int per_split = (nt_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t_begin = (int)blockIdx.y * per_split, t_end = min(nt_all, t_begin + per_split);
    if (gridDim.y > 1) ep.out = (float*)ep.out + (int64_t)blockIdx.y * M * ep.ldo;

    // per-lane global sources -------------------------------------------------
    // W: wave w stages fragment blocks c = 4w..4w+3 of the 16 (8 nb x 2 kb) in a stage
    const bf16_t* wsrc[4];
    int wdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = wave * 4 + i;
        const int nbl = c >> 1, kbl = c & 1;
        int nb = (n0 >> 4) + nbl;
        nb = min(nb, nblocks - 1);
        wsrc[i] = wp + ((int64_t)nb * kblocks + kbl) * 512 + lane * 8;
        wdst[i] = c * 1024;
    }
    // X: wave w stages row groups c = 4w..4w+3 (8 rows x 128 B each); XOR swizzle on the SOURCE chunk
    const bf16_t* xsrc[4];
    int xdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = wave * 4 + i;
        const int row = c * 8 + (lane >> 3);
        const int gch = (lane & 7) ^ (lane >> 3);      // (lds chunk) ^ (row & 7)
        int m = min(m0 + row, M - 1);
        xsrc[i] = x + (int64_t)m * ldx + gch * 8;
        xdst[i] = TILE_BYTES + c * 1024;
    }

    auto stage = [&](int t, int buf) {
        char* base = smem + buf * (2 * TILE_BYTES);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc[i] + (int64_t)t * 2 * 512), (lds_void*)(base + wdst[i]), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void*)(xsrc[i] + (int64_t)t * TK), (lds_void*)(base + xdst[i]), 16, 0, 0);
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets
    const int c16 = lane & 15, q4 = lane >> 4;
    int woff[4], xrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) woff[i] = ((wave_n * 4 + i) * 2) * 1024 + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) xrow[j] = wave_m * 64 + j * 16 + c16;

    if (t_begin < t_end) stage(t_begin, 0);
    int buf = 0;
    for (int t = t_begin; t < t_end; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < t_end) stage(t + 1, buf ^ 1);
        const char* wb = smem + buf * (2 * TILE_BYTES);
        const char* xb = wb + TILE_BYTES;
        // all 16 fragments of the K-tile are requested up front (one exposed LDS latency per tile instead of
        // one per 4-MFMA group); the compiler retires them with counted lgkmcnt waits as the MFMAs consume them
        bf16x8 wf[2][4], xf[2][4];
        auto ldx = [&](int kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = xrow[j];
                const int ch = (kk * 4 + q4) ^ (row & 7);
                xf[kk][j] = *(const bf16x8*)(xb + row * 128 + ch * 16);
            }
        };
        auto ldw = [&](int kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[kk][i] = *(const bf16x8*)(wb + woff[i] + kk * 1024);
        };
        auto mm = [&](int kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = mc_mfma_16x16x32(wf[kk][i], xf[kk][j], acc[i][j]);
        };
        // 12 reads in flight (lgkmcnt is a 4-bit counter), k-step 0 computes while k-step 1's fragments land
        ldx(0); ldw(0); ldx(1);
        __builtin_amdgcn_sched_barrier(0);
        ldw(1);
        mm(0);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        buf ^= 1;
    }

    // epilogue: acc[i][j][r] = out[m = ... + c16][n = ... + 4*q4 + r]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wave_m * 64 + j * 16 + c16;
        if (m >= M) continue;
        if (ep.swiglu) {
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const int n = n0 + wave_n * 64 + i * 16;
                if (n + 16 < N) epilogue_store4_swiglu(ep, m, (n >> 1) + q4 * 4, acc[i][j], acc[i + 1][j]);
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wave_n * 64 + i * 16 + q4 * 4;
            if (n < N) epilogue_store4(ep, m, n, acc[i][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// large-M kernel, 256x256x64 tiles, 8 waves in two groups that alternate LDS->register loads with MFMA work
// ------------------------------------------------------------------------------------------
// One workgroup (512 threads) per CU.  LDS = 2 K-tile buffers x (W 32 KiB | X 32 KiB) = 128 KiB, filled by
// LDS-DMA in 16-KiB "pieces" (2 x 1 KiB per wave):  W piece nh = the 64 weight rows each wave_n uses in output
// quadrant nh, X piece mh = the 32 tokens each wave_m uses in quadrant mh.
// A K-tile is 4 phases = the 4 output quadrants (nh,mh) in the order (0,0) (0,1) (1,1) (1,0); each phase is
//     [ds_read the operand half that changed | issue ONE piece of a later K-tile | counted vmcnt] s_barrier
//     [16 MFMA 16x16x32] s_barrier
// Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave feeds the matrix pipe while its partner
// loads.  Pieces are re-staged two phases after their last read (safe for both groups) and waited for with
// vmcnt(8): four younger pieces stay in flight across the barriers, nothing drains to zero in the main loop.
//   staging order per tile t:  P1 X1(t+1)  P2 W1(t+1)  P3 W0(t+2)  P4 X0(t+2)
//   waits (end of the load half): P1 X1(t)   P2 W1(t)   P4 W0(t+1),X0(t+1)
#define G2_STAGE 65536
#define G2_XOFF 32768

template <int N_>
__device__ __forceinline__ void g2_waitvm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}


// LDS-DMA of 16 bytes per lane, scalar-base form: wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset (one address VGPR
// instead of two, no v_lshl_add_u64 per piece); M0 = the wave-uniform LDS destination
__device__ __forceinline__ void g2_dma16s(const char* base_uniform, uint32_t off, uint32_t lds_addr_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base_uniform), "s"(lds_addr_uniform) : "memory");
}

struct G2Src {
    const char* wbase;       // wave-uniform bases (SGPRs); the K-tile offset is added to them on the scalar unit
    const char* xbase;
    uint32_t w[2][2];        // [nh][e]  byte offset of this lane's 16 bytes in K-tile 0
    uint32_t x[2][2];        // [mh][e]
};

// MODE 0: steady state (t <= nt-3), 1: t == nt-2, 2: t == nt-1
// ABL: timing-only ablations (bit 0 no LDS-DMA, bit 1 no fragment reads, bit 2 DMA always from K-tiles 0/1)
// Rejected placements of the two DMA instructions of a phase (measured): inside the MFMA half (-5 %), one in each
// half (-5 %); reading X0 of the next K-tile during P4 to balance the fragment reads 8/4/8/4 (+-0).
// NI: 16-column weight blocks per (wave_n, nh) quadrant: 4 = 256-column tiles, 3 = 192-column tiles (the fourth block slot of every LDS
// piece is then staged with a duplicate and never read, which keeps the DMA count per wave - and so the counted waits - unchanged)
template <int MODE, int ABL, int NI, int DIST = 2222, class LastHalf>
__device__ __forceinline__ void g2_tile(char* smem, int t, int wave, int woff, int xoff, const G2Src& src,
                                        f32x4 (&acc)[2][4][2][2], bf16x8 (&wf)[4][2], bf16x8 (&xf)[2][2][2], LastHalf&& last_half) {
    char* cur = smem + (t & 1) * G2_STAGE;
    char* nxt = smem + ((t + 1) & 1) * G2_STAGE;
    // piece ids: 0 = X1(t+1) -> nxt, 1 = W1(t+1) -> nxt, 2 = W0(t+2) -> cur, 3 = X0(t+2) -> cur
    auto stage1 = [&](int piece, int e) {
        if (ABL & 1) return;
        const bool on = (piece < 2) ? (MODE <= 1) : (MODE == 0);
        if (!on) return;
        int tt = (piece < 2) ? t + 1 : t + 2;
        if (ABL & 4) tt &= 1;
        char* buf = (piece < 2) ? nxt : cur;
        const bool is_w = (piece == 1 || piece == 2);
        const int h = (piece < 2) ? 1 : 0;
        // scalar-base form (round 3: +0.5 ... +2.5 % over per-lane 64-bit addresses): no vector instruction in the load halves but the DMA itself
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
        const uint32_t dst = lds0 + (uint32_t)(((piece < 2) ? ((t + 1) & 1) : (t & 1)) * G2_STAGE + (is_w ? 0 : G2_XOFF) + h * 16384 + (wave * 2 + e) * 1024);
        if (is_w) g2_dma16s(src.wbase + (int64_t)tt * 2048, src.w[h][e], dst);
        else g2_dma16s(src.xbase + (int64_t)tt * 128, src.x[h][e], dst);
    };
    // fragment reads in the order the MFMA cluster consumes them - k-step 0 of every block, then k-step 1 - pinned by sched_barriers (hipcc
    // otherwise issues them block-major and the cluster's first half waits for 11 of P1's 12 reads instead of 6)
    auto read_w1 = [&](int nh, int kk) {
        if ((ABL & 2) && t > 0) return;
#pragma unroll
        for (int i = 0; i < NI; ++i) wf[i][kk] = *(const bf16x8*)(cur + nh * 16384 + woff + i * 2048 + kk * 1024);
    };
    auto read_x1 = [&](int mh, int kk) {
        if ((ABL & 2) && t > 0) return;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) xf[mh][jj][kk] = *(const bf16x8*)(cur + G2_XOFF + mh * 16384 + ((xoff + jj * 2048) ^ (kk * 64)));
    };
    auto read_w = [&](int nh) {
        read_w1(nh, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_w1(nh, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_x = [&](int mh) {
        read_x1(mh, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_x1(mh, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_xw = [&](int mh, int nh) {          // P1: both operands
        read_x1(mh, 0);
        read_w1(nh, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_x1(mh, 1);
        read_w1(nh, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // load half: the phase's two DMA instructions, then the counted wait for the pieces the NEXT phase reads.
    // count = 2 x (pieces issued after the awaited one, before this point) = 8 in the steady state
    auto load_tail = [&](int piece, int wait_steady, int wait_m1, int wait_m2) {
        stage1(piece, 0);
        stage1(piece, 1);
        if (ABL & 1) return;
        const int w = MODE == 0 ? wait_steady : (MODE == 1 ? wait_m1 : wait_m2);
        switch (w) {
            case 0: g2_waitvm<0>(); break;
            case 2: g2_waitvm<2>(); break;
            case 4: g2_waitvm<4>(); break;
            case 7: g2_waitvm<7>(); break;
            case 8: g2_waitvm<8>(); break;
            default: break;        // -1: no wait
        }
    };
    // (s_setprio(1) around the MFMA cluster, as rounds 1-2 had it, costs 1.0-1.8 % on every LLM shape: ABL bit 6 builds it back for A/B;
    // raising the LOAD half's priority instead: +-0; DMA instructions moved behind the 12th / 16th MFMA of the preceding MFMA half: -1.5 ... -6 %)
    auto mma = [&](int nh, int mh) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((ABL & 64) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[nh][i][mh][jj] = mc_mfma_16x16x32(wf[i][kk], xf[mh][jj][kk], acc[nh][i][mh][jj]);
        if constexpr ((ABL & 64) != 0) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    if constexpr (DIST != 2222) {
        // Round 3: the tile's 8 DMA instructions a..h = X1(t+1) | W1(t+1) | W0(t+2) | X0(t+2) (issue order unchanged) dealt n1 / n2 / n3 / n4
        // over the four load halves (DIST = decimal n1 n2 n3 n4) instead of 2 / 2 / 2 / 2 - the halves carry 12 / 4 / 8 / 0 fragment reads, and
        // every distribution that keeps the DMA out of P1 measured +0.3 ... +1.6 % on the LLM shapes (0 / 2 / 3 / 3 ships for the 256-column
        // kernel; profiles/r03_probes/gemm_dma_distribution_ab.json; debug word (7 << 3) launches 2 / 2 / 2 / 2).  e..h overwrite operands of THIS tile's P1: they stay in P3 / P4 (n1 + n2 <= 4).  The
        // counted waits follow from the issue order: "X1(t) landed" = all but the 6 + n1 youngest, "W1(t)" = 4 + n1 + n2, "W0, X0(t+1)" = 8.
        constexpr int n1 = DIST / 1000, n2 = (DIST / 100) % 10, n3 = (DIST / 10) % 10, n4 = DIST % 10;
        static_assert(n1 + n2 + n3 + n4 == 8 && n1 + n2 <= 4, "bad DMA distribution");
        constexpr int m1 = n1 < 4 ? n1 : 4, m2 = (n1 + n2 < 4 ? n1 + n2 : 4) - m1;       // of a..d (the only ones tile nt-2 issues)
        auto issue = [&](int from, int to) {
#pragma unroll
            for (int i = from; i < to; ++i) stage1(i >> 1, i & 1);
        };
        auto tail_wait = [&](auto ws, auto w1, auto w2) {
            if (ABL & 1) return;
            if constexpr (MODE == 0) g2_waitvm<decltype(ws)::value>();
            else if constexpr (MODE == 1) g2_waitvm<decltype(w1)::value>();
            else if constexpr (decltype(w2)::value >= 0) g2_waitvm<decltype(w2)::value>();
        };
        read_xw(0, 0);
        issue(0, n1);
        tail_wait(std::integral_constant<int, 6 + n1>{}, std::integral_constant<int, 6 + m1>{}, std::integral_constant<int, 2>{});
        mma(0, 0);
        read_x(1);
        issue(n1, n1 + n2);
        tail_wait(std::integral_constant<int, 4 + n1 + n2>{}, std::integral_constant<int, 4 + m1 + m2>{}, std::integral_constant<int, 0>{});
        mma(0, 1);
        read_w(1);
        issue(n1 + n2, n1 + n2 + n3);
        mma(1, 1);
        issue(n1 + n2 + n3, 8);
        tail_wait(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, -1>{});
        if constexpr (MODE == 2) last_half();          // the launch's last load half: nothing is staged, nothing waited for
        mma(1, 0);
        return;
    }
    // P1 (0,0): reads W0,X0; stages X1(t+1); waits for X1(t)
    read_xw(0, 0);
    load_tail(0, 8, 8, 2);
    mma(0, 0);
    // P2 (0,1): reads X1; stages W1(t+1); waits for W1(t)
    read_x(1);
    load_tail(1, 8, 8, 0);
    mma(0, 1);
    // P3 (1,1): reads W1; stages W0(t+2); nothing to wait for
    read_w(1);
    load_tail(2, -1, -1, -1);
    mma(1, 1);
    // P4 (1,0): stages X0(t+2); waits for W0(t+1), X0(t+1)
    load_tail(3, 8, 4, -1);
    if constexpr (MODE == 2) last_half();
    mma(1, 0);
}

// (Round 4, measured and NOT in the tree - profiles/r04_probes/gemm_store_sc1_ab.json: the epilogue's 16-byte output stores as `sc1` / `sc0 sc1`
// (write through, line dropped from the XCD's L2, so that a round's 4 MiB of output per XCD does not pass through the cache that holds the
// operand slices): -17 ... -43 % on every shape - the write-back L2 is what absorbs the 128 KiB per tile; residual rows by nontemporal
// loads: -0.1 ... -0.9 % at K >= 4096, -9 % at K = 1024.)
// wide epilogue of the 256x256 kernel for plain bf16 outputs (bias / activation / residual), one instantiation per activation class so
// that no element walks a branch tree: every lane stores 16 bytes per block pair (v_permlane16_swap, see above)
template <int ACTC, int BIASC, int RESC, int NI, bool SS = false>
__device__ __forceinline__ void g2_epilogue_wide(const Epilogue& ep, f32x4 (&acc)[2][4][2][2], bf16x4 (&res)[2][2][2][NI], bool has_res, int m0,
                                                 int n0, int M, int wave_m, int wave_n, int c16, int q4, float act_k) {
    const int nw = n0 + wave_n * (2 * NI * 16);
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int m = m0 + wave_m * 64 + mh * 32 + jj * 16 + c16;
            const bool live = m < M;                   // depends on c16 only: the lane pairs of a swap are live together
            const int mc = live ? m : (M - 1);
            const float a = ep.row_scale ? ep.alpha * ep.row_scale[mc] : ep.alpha;
            bf16_t* orow = (bf16_t*)ep.out + (int64_t)mc * ep.ldo + nw;
            float ss = 0.f;
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int i = 0; i < NI; i += 2) {
                    const int nb = nh * (NI * 16) + i * 16;
                    const bf16x4 lo = epilogue_vals4<ACTC, BIASC, RESC>(ep, a, nw + nb + q4 * 4, acc[nh][i][mh][jj], has_res, res[mh][jj][nh][i], act_k);
                    const bf16x4 hi = epilogue_vals4<ACTC, BIASC, RESC>(ep, a, nw + nb + 16 + q4 * 4, acc[nh][i + 1][mh][jj], has_res, res[mh][jj][nh][i + 1], act_k);
                    if constexpr (SS) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float f0 = (float)lo[e], f1 = (float)hi[e];
                            ss = fmaf(f0, f0, ss);
                            ss = fmaf(f1, f1, ss);
                        }
                    }
                    u32x2 pa = __builtin_bit_cast(u32x2, lo), pb = __builtin_bit_cast(u32x2, hi);
                    auto r0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
                    auto r1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
                    const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                    if (live) *(u32x4*)(orow + nb + (q4 & 1) * 16 + (q4 >> 1) * 8) = o;
                }
            if constexpr (SS) {
                // this wave's 128 columns of row m: the four column quarters (q4) of the row, then one store per row
                ss += __shfl_xor(ss, 16, 64);
                ss += __shfl_xor(ss, 32, 64);
                if (live && q4 == 0) ep.ss_parts[(int64_t)m * ep.ss_chunks + (nw >> 7)] = ss;
            }
        }
}

// rows of one launch may belong to several adapter groups (routed LocalLoRA order): group g owns rows [row_start[g], row_start[g+1])
// = m-tiles [tile_start[g], tile_start[g+1]) and multiplies against its own composed weight
struct G2Groups {
    int n;
    int tile_start[9];
    int row_start[9];
    const bf16_t* wp[8];
};

// adapter group of m-tile tm: first row, end row, weight - running selects over statically indexed kernel arguments (a dynamic index
// into the by-value struct costs a dependent s_load per field between the block id and the first DMA)
__device__ __forceinline__ void g2_group_of(const G2Groups& grp, int tm, int& m0, int& M, const bf16_t*& wp) {
    int rs = grp.row_start[0], ts = grp.tile_start[0], re = grp.row_start[1];
    wp = grp.wp[0];
#pragma unroll
    for (int i = 1; i < 8; ++i)
        if (i < grp.n && tm >= grp.tile_start[i]) { rs = grp.row_start[i]; ts = grp.tile_start[i]; re = grp.row_start[i + 1]; wp = grp.wp[i]; }
    m0 = rs + (tm - ts) * 256;
    M = re;
}

// blockIdx -> (tm, tn) of gemm_tile256_kernel
__device__ __forceinline__ void g2_map_tile(int bid, int nwg, int tiles_m, int tiles_n, int raster, int& tm, int& tn) {
    if ((raster & 255) == 1) {
        const int full = nwg & ~255;
        if (bid < full) {
            const int xcd = bid & 7, idx = bid >> 3;
            bid = (((idx >> 5) << 3) + xcd) * 32 + (idx & 31);
        }
    } else {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int GROUP = 8;
    // raster bits 8-15 = SW > 0 (round 3): tile columns in SLABS of SW, slab outermost - all m-groups of a slab before the next slab - so
    // that the slab of W (SW x 256 x K x 2 bytes: 67 MB at SW = 32, K = 4096) stays in the 256-MiB Infinity Cache across the whole sweep
    // over M instead of the whole W being re-swept (and, once the activations + outputs exceed the cache, re-fetched from HBM) once per
    // m-group; with SW = 32 an (m-group, slab) is exactly one 8 x 4 block per XCD
    int tn_base = 0, tiles_n_eff = tiles_n;
    const int SW = (raster >> 8) & 255;
    if (SW > 0) {
        const int per_slab = tiles_m * SW;
        const int sidx = bid / per_slab;
        tn_base = sidx * SW;
        tiles_n_eff = min(SW, tiles_n - tn_base);
        bid -= sidx * per_slab;
    }
    // full m-groups (all but possibly the last) need one division, by tiles_n_eff; m fastest inside a group
    const int full_groups = tiles_m >> 3;
    const int g8 = (bid >> 3) / tiles_n_eff;      // group index if every group before this tile is full
    if (g8 < full_groups) {
        const int in_group = bid - g8 * (GROUP * tiles_n_eff);
        tm = g8 * GROUP + (in_group & 7);
        tn = tn_base + (in_group >> 3);
        return;
    }
    const int first_m = full_groups * GROUP;
    const int gsz = tiles_m - first_m;            // 1 .. 7 m-tiles in the last group
    const int rest = bid - full_groups * (GROUP * tiles_n_eff);
    tm = first_m + rest % gsz;
    tn = tn_base + rest / gsz;
}

// in-kernel clock diagnostic (ABL bit 3): shader-clock and 100 MHz real-time ticks across one workgroup's main loop
__device__ unsigned long long g2_stamps[2 * 4096];

template <int ABL, int NI = 4, int DIST = 2222>
__global__ __launch_bounds__(512, (ABL & 16) ? 1 : 2) void gemm_tile256_kernel(const bf16_t* __restrict__ x, int64_t ldx, G2Groups grp, int N, int K,
                                                              Epilogue ep, int tiles_m, int tiles_n, int raster) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: LDS-DMA destinations stay on the scalar unit
    const int wave_n = wave >> 2, wave_m = wave & 3;
    const int c16 = lane & 15, q4 = lane >> 4;

    // XCD-aware tile order.  Logical order: groups of 8 m-tiles, inside a group n-tile columns, m fastest, so that 32 consecutive
    // tiles are an 8 x 4 block (12 tile-rows of operands for 32 tiles through one XCD's L2).  Blocks b and b+8 of the grid share an XCD
    // (round-robin dispatch; speed only).  raster 0: every XCD owns a contiguous eighth of the logical order - the XCDs work on
    // different m-groups, 8 groups' activations (8 x 256 x K each) plus the weight sweep are live in the 256-MiB Infinity Cache.
    // raster 1 (large M): the 32-tile blocks are dealt round-robin over the XCDs, so all eight work on the SAME m-group at the same
    // time: its activations are fetched from HBM once and hit the Infinity Cache for the other seven, and the weight matrix, re-swept
    // once per m-group, stays resident - L2 misses become Infinity-Cache hits instead of HBM round trips.
    const int nwg = tiles_m * tiles_n;
    int tm, tn;
    g2_map_tile(blockIdx.x, nwg, tiles_m, tiles_n, raster, tm, tn);
    constexpr int NT = NI * 64;                   // tile width in weight rows: 2 wave_n x 2 nh x NI blocks of 16
    int m0, M;                                    // rows of this tile beyond the group's end (M) are clamped on load and not stored
    const bf16_t* wp;
    g2_group_of(grp, tm, m0, M, wp);
    const int n0 = tn * NT;

    const int kblocks = K >> 5;
    const int nblocks = (N + 15) >> 4;
    const int nt = K >> 6;

    const char* pf_w = nullptr;
    const char* pf_x = nullptr;
    // L2 warm-up for the workgroup that follows this one on the CU (round 3; ABL bit 8 builds it out for A/B).  One workgroup per CU and equal
    // tile times: block b + 256 starts where block b ends, on the same XCD (b mod 8).  Its prologue waits for K-tiles 0 and 1 of its
    // operands (128 KiB = 1024 lines, 2 - 3 us from the Infinity Cache / HBM with nothing else to do); this workgroup touches those lines
    // from its LAST load half - no DMA is issued there and nothing waits on the vector-memory counter any more - as LDS-DMA into a
    // scratch KiB nobody reads: one line per lane, two instructions per wave.  +0.8 ... +2.2 % at K >= 4096 (q|k|v +1.0, o +1.7,
    // gate|up +1.2, down +1.6); at K = 1024 the extra fills cost more than the shorter prologue gives (-0.4 ... -1.5 %): from 32 K-tiles up.
    // (K-tiles 0 .. 3 instead of 0, 1: no better)
    if constexpr ((ABL & 256) == 0) {
        const int nb_ = (int)blockIdx.x + 256;
        if (nb_ < nwg && nt >= 32) {
            int tm2, tn2;
            g2_map_tile(nb_, nwg, tiles_m, tiles_n, raster, tm2, tn2);
            int m02, M2;
            const bf16_t* wp2;
            g2_group_of(grp, tm2, m02, M2, wp2);
            const int nb2 = min(((tn2 * NT) >> 4) + (tid >> 5), nblocks - 1);          // 16 weight block-rows x 32 lines (K-tiles 0, 1 = 4 KiB each)
            pf_w = (const char*)wp2 + (int64_t)nb2 * kblocks * 1024 + (tid & 31) * 128;
            pf_x = (const char*)x + (int64_t)min(m02 + (tid >> 1), M2 - 1) * ldx * 2 + (tid & 1) * 128;      // 256 rows x 2 lines
        }
    }
    auto last_half = [&] {
        if constexpr ((ABL & 256) == 0) {
            if (pf_w) {
                const uint32_t scratch = (uint32_t)(uintptr_t)(lds_void*)smem + 2 * G2_STAGE;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\tglobal_load_lds_dword %1, off" ::"v"(pf_w), "v"(pf_x), "s"(scratch) : "memory");
            }
        }
    };

    G2Src src;
    src.wbase = (const char*)wp;
    src.xbase = (const char*)x;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = wave * 2 + e;
            // W: block c of piece h  ->  (wn, i, kk)
            const int wn = c >> 3, i = (c >> 1) & 3, kk = c & 1;
            const int nb = min((n0 >> 4) + wn * (2 * NI) + h * NI + min(i, NI - 1), nblocks - 1);
            src.w[h][e] = (uint32_t)((((int64_t)nb * kblocks + kk) * 512 + lane * 8) * 2);
            // X: block c of piece h  ->  8 rows of wave_m group wm
            const int wm = c >> 2, r8 = c & 3;
            const int row = wm * 64 + h * 32 + r8 * 8 + (lane >> 3);
            const int gch = (lane & 7) ^ (lane >> 3);
            src.x[h][e] = (uint32_t)(((int64_t)min(m0 + row, M - 1) * ldx + gch * 8) * 2);
        }
    const int woff = wave_n * 8192 + lane * 16;
    const int xoff = wave_m * 4096 + c16 * 128 + ((q4 ^ (c16 & 7)) * 16);

    f32x4 acc[2][4][2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 2; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: all of K-tile 0, then W0 / X0 of K-tile 1 (the order the steady state would have produced)
    {
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)smem;
        auto pw = [&](int buf, int nh, int tt) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                g2_dma16s(src.wbase + (int64_t)tt * 2048, src.w[nh][e], lds0 + (uint32_t)(buf * G2_STAGE + nh * 16384 + (wave * 2 + e) * 1024));
        };
        auto px = [&](int buf, int mh, int tt) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                g2_dma16s(src.xbase + (int64_t)tt * 128, src.x[mh][e], lds0 + (uint32_t)(buf * G2_STAGE + G2_XOFF + mh * 16384 + (wave * 2 + e) * 1024));
        };
        pw(0, 0, 0); px(0, 0, 0); px(0, 1, 0); pw(0, 1, 0); pw(1, 0, 1); px(1, 0, 1);
        g2_waitvm<8>();
        __builtin_amdgcn_s_barrier();
    }
    bf16x8 wf[4][2], xf[2][2][2];
    // Every scalar load of the prologue (kernel arguments) is retired HERE, explicitly.  hipcc's wait-count pass otherwise carries "a
    // scalar load may still be pending" into the loop header, and because scalar loads return out of order it then guards the first MFMA
    // cluster of every K-tile with one lgkmcnt(0) instead of the counted lgkmcnt(7) / (5) / (3) / (1) that let the MFMAs start behind the
    // first five of P1's twelve fragment reads - measured -1.5 ... -2.3 % on the K >= 4096 shapes, and which of the two it emits flips with
    // unrelated changes to the prologue (profiles/r03_probes/README.md)
    __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0), vmcnt / expcnt untouched
    if (wave_n == 1) __builtin_amdgcn_s_barrier();      // second group runs one barrier behind
    int t = 0;
    unsigned long long st0 = 0, sr0 = 0;
    if (ABL & 8) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
    for (; t < nt - 2; ++t) g2_tile<0, ABL, NI, DIST>(smem, t, wave, woff, xoff, src, acc, wf, xf, last_half);
    g2_tile<1, ABL, NI, DIST>(smem, t, wave, woff, xoff, src, acc, wf, xf, last_half);
    g2_tile<2, ABL, NI, DIST>(smem, t + 1, wave, woff, xoff, src, acc, wf, xf, last_half);
    if (ABL & 8) {
        const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && blockIdx.x < 4096) { g2_stamps[2 * blockIdx.x] = st1 - st0; g2_stamps[2 * blockIdx.x + 1] = sr1 - sr0; }
    }
    if (wave_n == 0) __builtin_amdgcn_s_barrier();
    if constexpr ((ABL & 32) != 0) {           // timing-only (tools/probes/gemm_fixed_cost_probe.py): no epilogue; one never-taken store keeps the accumulators live
        float t_ = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) t_ += acc[a][b][c][d][0] + acc[a][b][c][d][1] + acc[a][b][c][d][2] + acc[a][b][c][d][3];
        if (t_ == 12345.678f) ((float*)ep.out)[0] = t_;
        return;
    }

    // Epilogue.  bf16 outputs whose tile lies inside N take the wide path (16-byte stores of block pairs, the row factor loaded once per
    // row, ALL residual values of the lane requested before the first store: one exposed memory latency per tile instead of one per row -
    // a workgroup owns its CU, so nothing else runs while its epilogue waits); everything else the 8-byte path.
    const bool wide = !ep.out_f32 && (n0 + NT <= N) && (ep.ldo % 8 == 0) && ((uintptr_t)ep.out % 16 == 0) && (NI % 2 == 0);
    if constexpr (NI == 4) {
        if (wide && ep.rope.q_out) {
            // RoPE + scatter (what rope_kv_kernel does to the stored q|k|v row): the values are rounded to bf16 first, exactly as the unfused
            // route stores them, then rotated in fp32 and rounded again.  A wave's 128 columns are one head.
            const Epilogue::Rope& rp = ep.rope;
            const int nw = n0 + wave_n * 128;
            const int head = nw >> 7;
            const bool rot = head < rp.H + rp.Hkv;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int m = m0 + wave_m * 64 + mh * 32 + jj * 16 + c16;
                    const bool live = m < M;
                    const int mc = live ? m : (M - 1);
                    const float a = ep.row_scale ? ep.alpha * ep.row_scale[mc] : ep.alpha;
                    const int b = rp.row_b[mc], pos = rp.row_pos[mc], tq = rp.row_t[mc];
                    const bool put = live && b >= 0;
                    bf16_t* drow;
                    if (head < rp.H) drow = rp.q_out + ((int64_t)(b * rp.Lq + tq) * rp.H + head) * 128;
                    else if (rot) drow = rp.k_cache + (((int64_t)b * rp.Hkv + (head - rp.H)) * rp.Smax + pos) * 128;
                    else drow = rp.v_cache + (((int64_t)b * rp.Hkv + (head - rp.H - rp.Hkv)) * rp.Smax + pos) * 128;
                    const float* cr = rp.cosT + (int64_t)pos * 64 + q4 * 4;
                    const float* sr = rp.sinT + (int64_t)pos * 64 + q4 * 4;
                    bf16x4 v1[4], v2[4];                   // block i: first-half / second-half values of d = 16 i + 4 q4 ..
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v1[i] = epilogue_vals4<0, -1, 0>(ep, a, nw + i * 16 + q4 * 4, acc[0][i][mh][jj], false, (bf16x4){0, 0, 0, 0});
                        v2[i] = epilogue_vals4<0, -1, 0>(ep, a, nw + 64 + i * 16 + q4 * 4, acc[1][i][mh][jj], false, (bf16x4){0, 0, 0, 0});
                    }
                    if (rot) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 c = *(const f32x4*)(cr + i * 16), sn = *(const f32x4*)(sr + i * 16);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float r1, r2;
                                mc_rope_pair((float)v1[i][j], (float)v2[i][j], c[j], sn[j], r1, r2);
                                v1[i][j] = (bf16_t)r1; v2[i][j] = (bf16_t)r2;
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {
                        u32x2 pa = __builtin_bit_cast(u32x2, v1[i]), pb = __builtin_bit_cast(u32x2, v1[i + 1]);
                        auto r0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
                        auto r1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
                        const u32x4 o1 = {r0[0], r1[0], r0[1], r1[1]};
                        if (put) *(u32x4*)(drow + i * 16 + (q4 & 1) * 16 + (q4 >> 1) * 8) = o1;
                        pa = __builtin_bit_cast(u32x2, v2[i]); pb = __builtin_bit_cast(u32x2, v2[i + 1]);
                        r0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
                        r1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
                        const u32x4 o2 = {r0[0], r1[0], r0[1], r1[1]};
                        if (put) *(u32x4*)(drow + 64 + i * 16 + (q4 & 1) * 16 + (q4 >> 1) * 8) = o2;
                    }
                }
            return;
        }
    }
    const bool has_res = wide && !ep.swiglu && ep.residual != nullptr;
    const int actc = ep.act == MC_ACT_NONE ? 0 : ((ep.act == MC_ACT_QUICK_GELU || ep.act == MC_ACT_SILU) ? 1 : (ep.act == MC_ACT_GELU ? 3 : 2));
    const float act_k = ep.act == MC_ACT_QUICK_GELU ? 1.702f : 1.0f;
    bf16x4 res[2][2][2][NI];
    // residual rows in 16-byte loads (round 3): the lane reads the 8 columns it will STORE (the layout behind the epilogue's
    // v_permlane16_swap) and swaps them back into the accumulator layout - the exchange is its own inverse; half the load instructions
    const bool res16 = has_res && NI % 2 == 0 && ep.ldr % 8 == 0 && ((uintptr_t)ep.residual % 16 == 0) && !(ABL & 1024);
    if (res16) {
        if constexpr (NI % 2 == 0) {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int mc = min(m0 + wave_m * 64 + mh * 32 + jj * 16 + c16, M - 1);
                    const bf16_t* rrow = ep.residual + (int64_t)mc * ep.ldr + n0 + wave_n * (2 * NI * 16) + (q4 & 1) * 16 + (q4 >> 1) * 8;
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int i = 0; i < NI; i += 2) {
                            const u32x4 v = *(const u32x4*)(rrow + nh * (NI * 16) + i * 16);
                            auto r0 = __builtin_amdgcn_permlane16_swap(v[0], v[2], false, false);
                            auto r1 = __builtin_amdgcn_permlane16_swap(v[1], v[3], false, false);
                            const u32x2 lo = {r0[0], r1[0]}, hi = {r0[1], r1[1]};
                            res[mh][jj][nh][i] = __builtin_bit_cast(bf16x4, lo);
                            res[mh][jj][nh][i + 1] = __builtin_bit_cast(bf16x4, hi);
                        }
                }
        }
    } else if (has_res) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int mc = min(m0 + wave_m * 64 + mh * 32 + jj * 16 + c16, M - 1);
                const bf16_t* rrow = ep.residual + (int64_t)mc * ep.ldr + n0 + wave_n * (2 * NI * 16) + q4 * 4;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int i = 0; i < NI; ++i) res[mh][jj][nh][i] = *(const bf16x4*)(rrow + nh * (NI * 16) + i * 16);
            }
    }
    if constexpr (NI % 2 == 0) {
        if (wide && !ep.swiglu) {
            // the common combinations get their own straight-line instantiation; the rest decide bias / residual per call
#define G2_EPI(A, B, R) g2_epilogue_wide<A, B, R, NI>(ep, acc, res, has_res, m0, n0, M, wave_m, wave_n, c16, q4, act_k)
            const bool hb = ep.bias != nullptr;
            if (ep.ss_parts && NI == 4 && actc == 0 && !hb && has_res) g2_epilogue_wide<0, 0, 1, NI, true>(ep, acc, res, has_res, m0, n0, M, wave_m, wave_n, c16, q4, act_k);   // LLM o / down + next norm's factor
            else if (ep.ss_parts && NI == 4 && actc == 0) g2_epilogue_wide<0, -1, -1, NI, true>(ep, acc, res, has_res, m0, n0, M, wave_m, wave_n, c16, q4, act_k);
            else if (actc == 0 && !hb && !has_res) G2_EPI(0, 0, 0);            // LLM q|k|v (without the RoPE route), plain projections
            else if (actc == 0 && !hb && has_res) G2_EPI(0, 0, 1);        // LLM o / down
            else if (actc == 0 && hb && !has_res) G2_EPI(0, 1, 0);        // encoder q|k|v
            else if (actc == 0 && hb && has_res) G2_EPI(0, 1, 1);         // encoder out / fc2
            else if (actc == 1 && hb && !has_res) G2_EPI(1, 1, 0);        // encoder fc1 (QuickGELU)
            else if (actc == 1) G2_EPI(1, -1, -1);
            else if (actc == 3) G2_EPI(3, -1, -1);                        // BEATs / point-cloud fc1 (exact GELU)
            else G2_EPI(2, -1, -1);
#undef G2_EPI
            return;
        }
    }
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int m = m0 + wave_m * 64 + mh * 32 + jj * 16 + c16;
            const bool live = m < M;                   // depends on c16 only: the lane pairs of a swap are live together
            const int mc = live ? m : (M - 1);
            if (wide) {
                const float a = ep.row_scale ? ep.alpha * ep.row_scale[mc] : ep.alpha;
                if (ep.swiglu) {
                    // gate / up blocks alternate along N: input blocks (f, f+1) -> output block f/2; output blocks f/2 and f/2+1 are adjacent
                    bf16_t* orow = (bf16_t*)ep.out + (int64_t)mc * ep.ldo + ((n0 + wave_n * (2 * NI * 16)) >> 1);
#pragma unroll
                    for (int f = 0; f < 2 * NI; f += 4) {
                        const bf16x4 lo = swiglu_vals4(a, acc[f / NI][f % NI][mh][jj], acc[(f + 1) / NI][(f + 1) % NI][mh][jj]);
                        const bf16x4 hi = swiglu_vals4(a, acc[(f + 2) / NI][(f + 2) % NI][mh][jj], acc[(f + 3) / NI][(f + 3) % NI][mh][jj]);
                        u32x2 pa = __builtin_bit_cast(u32x2, lo), pb = __builtin_bit_cast(u32x2, hi);
                        auto r0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
                        auto r1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
                        const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                        if (live) *(u32x4*)(orow + (f >> 1) * 16 + (q4 & 1) * 16 + (q4 >> 1) * 8) = o;
                    }
                    continue;
                }
                const int nw = n0 + wave_n * (2 * NI * 16);
                bf16_t* orow = (bf16_t*)ep.out + (int64_t)mc * ep.ldo + nw;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int i = 0; i < NI; i += 2) {
                        const int nb = nh * (NI * 16) + i * 16;
                        const bf16x4 lo = epilogue_vals4(ep, a, nw + nb + q4 * 4, acc[nh][i][mh][jj], has_res, res[mh][jj][nh][i]);
                        const bf16x4 hi = epilogue_vals4(ep, a, nw + nb + 16 + q4 * 4, acc[nh][i + 1][mh][jj], has_res, res[mh][jj][nh][i + 1]);
                        u32x2 pa = __builtin_bit_cast(u32x2, lo), pb = __builtin_bit_cast(u32x2, hi);
                        auto r0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
                        auto r1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
                        const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
                        if (live) *(u32x4*)(orow + nb + (q4 & 1) * 16 + (q4 >> 1) * 8) = o;
                    }
                continue;
            }
            if (!live) continue;
            if (ep.swiglu) {
                // gate / up blocks alternate along N: consecutive block pairs of this wave's 2*NI blocks (a pair may straddle the nh halves
                // when NI is odd; both halves are this lane's registers)
#pragma unroll
                for (int f = 0; f < 2 * NI; f += 2) {
                    const int n = n0 + wave_n * (2 * NI * 16) + f * 16;
                    if (n + 16 < N) epilogue_store4_swiglu(ep, m, (n >> 1) + q4 * 4, acc[f / NI][f % NI][mh][jj], acc[(f + 1) / NI][(f + 1) % NI][mh][jj]);
                }
                continue;
            }
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int n = n0 + wave_n * (2 * NI * 16) + nh * (NI * 16) + i * 16 + q4 * 4;
                    if (n < N) epilogue_store4(ep, m, n, acc[nh][i][mh][jj]);
                }
        }
}

// Split-K second stage for the 128x128 kernel: sums the fp32 slabs [S][M][N] in fixed order and applies the real epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int S, int M, int N, Epilogue ep) {
    const int nq = N >> 2;
    const int64_t total = (int64_t)M * nq;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / nq), n = (int)(i % nq) * 4;
        f32x4 v = *(const f32x4*)(slabs + (int64_t)m * N + n);
        for (int k = 1; k < S; ++k) v += *(const f32x4*)(slabs + ((int64_t)k * M + m) * N + n);
        epilogue_store4(ep, m, n, v);
    }
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>
// Optional live timing of the dominant kernel (gemm_tile256_kernel): HIP events recorded on the launch stream around every launch while
// enabled (bench.py's roofline object).  Not graph-capturable; leave disabled in normal operation.
namespace {
struct ProfRec { hipEvent_t a, b; double flops, bytes; int K; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
}  // namespace

// Diagnostic word (kernel A/B builds, forced tile shapes): only in the probes build of the library (make probes -> tools/probes/
// libmc_hip_probes.so, -DMC_PROBES); the shipped library has neither the entry point nor the variants it selects.
#ifdef MC_PROBES
static int g_gemm_dbg = 0;
extern "C" int mc_gemm_debug(int v) { g_gemm_dbg = v; return 0; }
#else
constexpr int g_gemm_dbg = 0;
#endif

// library-owned scratch for the automatic split-K of under-filled 128x128 launches (one process drives one GPU and one compute stream; the
// buffer only grows, and growing synchronises the device first so no launch in flight still reads the old one)
static float* splitk_workspace(size_t floats) {
    static float* buf = nullptr;
    static size_t cap = 0;
    if (floats > cap) {
        (void)hipDeviceSynchronize();
        if (buf) (void)hipFree(buf);
        buf = nullptr; cap = 0;
        if (hipMalloc((void**)&buf, floats * sizeof(float)) != hipSuccess) { buf = nullptr; return nullptr; }
        cap = floats;
    }
    return buf;
}

// Tile width of the large-M kernel for m_tiles row tiles: 256 columns, or 192 when that fills the 256 CUs better.  One workgroup per CU,
// so a launch takes ceil(tiles / 256) rounds; a 192-column tile does 3/4 of the work of a 256-column one at ~0.92 of its MFMA efficiency
// (12 instead of 16 MFMAs per phase behind the same X reads and DMA issues).  M = 2728 (the finetune step), N = 4096: 176 tiles of 256
// fill 69 % of the CUs, 242 tiles of 192 fill 95 %.  (probes build: debug word bit 10 forces 192, bit 11 forces 256)
// "tile192" = 0 keeps the large-M kernel on 256-column tiles: for callers that fill the idle CUs of an under-filled launch themselves
// (the finetune step runs its rank-projection and weight-gradient GEMMs on a second stream next to the base GEMMs: measured on one
// device, samples/s  overlap + 256: 33.4, overlap + 192: 31.8, no overlap + 192: 32.8, no overlap + 256: 31.1).  The one policy knob of
// the shipped library; it does not change results.
static bool g_tile192 = true;
// Raster of the 256-row kernel (measured defaults, rounds 3-4; the probes build can move them):
static bool g_raster_auto = true;          // launches of >= g_raster_min_tiles tiles deal their 32-tile blocks round-robin over the XCDs
static int g_raster_slab = 32;             // tile columns per n-slab of the shared-m-group raster (0 = no slabs)
static int g_raster_slab_min = 64;         // ... for launches of more than this many tile columns (gate|up: 86; measured, M = 44 656, K = 4096:
                                           // N = 22 016 1398 -> 1425 TFLOP/s, N = 12 288 1437 -> 1425: q|k|v keeps the slab-less order)
static int g_raster_min_tiles = 1024;
extern "C" int mc_gemm_set_option(const char* name, int value) {
    if (name && !strcmp(name, "tile192")) { g_tile192 = value != 0; return 0; }
#ifdef MC_PROBES
    if (name && !strcmp(name, "raster_shared")) { g_raster_auto = value != 0; return 0; }
    if (name && !strcmp(name, "raster_slab")) { g_raster_slab = value < 0 ? 0 : (value > 255 ? 255 : value); g_raster_slab_min = 0; return 0; }
    if (name && !strcmp(name, "raster_min_tiles")) { g_raster_min_tiles = value; return 0; }
#endif
    mc_set_error("mc_gemm_set_option: unknown option '%s'", name ? name : "(null)");
    return 1;
}

static int tile_ni(int64_t m_tiles, int N, bool swiglu) {
    if (g_gemm_dbg & 2048) return 4;
    if (!g_tile192 && !(g_gemm_dbg & 1024)) return 4;
    if (swiglu) return 4;
    if (g_gemm_dbg & 1024) return 3;
    const int64_t t256 = m_tiles * ((N + 255) / 256), t192 = m_tiles * ((N + 191) / 192);
    const double e256 = (double)t256 / (256.0 * ((t256 + 255) / 256));
    const double e192 = 0.92 * (double)t192 / (256.0 * ((t192 + 255) / 256));
    return e192 > e256 * 1.05 ? 3 : 4;
}

// 256-row tiles need enough tiles to fill most of the 256 CUs; bit 1 of the debug word forces the 128x128 kernel, bit 2 the 256-row one
static bool use_tile256(int M, int N, int K) {
    if (K < 128) return false;
    if (g_gemm_dbg & 2) return false;
    if (g_gemm_dbg & 4) return true;
    // measured crossover on MI355X (tools/bench_ops.py mid): 176 tiles (M=2732, N=4096) 1.15-1.4x faster than the 128x128 kernel,
    // 112 tiles (M=1552) 1.3-1.4x slower
    const int64_t mt = (M + 255) / 256;
    const int64_t tiles = mt * ((N + (tile_ni(mt, N, false) == 3 ? 191 : 255)) / (tile_ni(mt, N, false) == 3 ? 192 : 256));
    return tiles >= 144;
}

extern "C" int mc_gemm_profile_enable(int on) {
    if (on && !g_prof_on) {
        for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        g_prof.clear();
    }
    g_prof_on = on != 0;
    return 0;
}

// total elapsed (ms), total algorithmic flops and launch count of gemm_tile256_kernel since the last enable
extern "C" int mc_gemm_profile_read(double* total_ms, double* total_flops, int64_t* launches) {
    double ms = 0.0, fl = 0.0;
    for (auto& r : g_prof) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) { mc_set_error("mc_gemm_profile_read: %s", hipGetErrorString(e)); return 2; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        ms += t; fl += r.flops;
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (launches) *launches = (int64_t)g_prof.size();
    return 0;
}

#ifdef MC_PROBES
// Diagnostic (debug word 40 must have been set for the launches): median over the first n_wg workgroups of the last 256x256 launch of
// the shader clock held across the main loop, in GHz (delta s_memtime / delta s_memrealtime x 100 MHz).
extern "C" int mc_gemm_clock_read(int n_wg, double* ghz) {
    MC_CHECK_ARG(ghz && n_wg > 0 && n_wg <= 4096, "mc_gemm_clock_read: bad arguments");
    std::vector<unsigned long long> h(2 * (size_t)n_wg);
    hipError_t e = hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g2_stamps), h.size() * sizeof(unsigned long long));
    if (e != hipSuccess) { mc_set_error("mc_gemm_clock_read: %s", hipGetErrorString(e)); return 2; }
    std::vector<double> r;
    for (int i = 0; i < n_wg; ++i)
        if (h[2 * i + 1]) r.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    if (r.empty()) { mc_set_error("mc_gemm_clock_read: no stamps recorded"); return 2; }
    std::sort(r.begin(), r.end());
    *ghz = r[r.size() / 2];
    return 0;
}
#endif

// the same sums over the launches with k_min <= K <= k_max only (bench.py: the decoder layers' launches, K >= 4096, apart from the encoder
// towers', K <= 1024, whose per-tile prologue + epilogue weigh 4x more)
extern "C" int mc_gemm_profile_read_range(int k_min, int k_max, double* total_ms, double* total_flops, int64_t* launches) {
    double ms = 0.0, fl = 0.0;
    int64_t n = 0;
    for (auto& r : g_prof) {
        if (r.K < k_min || r.K > k_max) continue;
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) { mc_set_error("mc_gemm_profile_read_range: %s", hipGetErrorString(e)); return 2; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        ms += t; fl += r.flops; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (launches) *launches = n;
    return 0;
}

extern "C" int mc_gemm_profile_read_bytes(double* total_bytes) {
    double by = 0.0;
    for (auto& r : g_prof) by += r.bytes;
    if (total_bytes) *total_bytes = by;
    return 0;
}

extern "C" int mc_packed_weight_elems(int N, int K, int64_t* out_elems) {
    const int64_t Np = (N + 15) / 16 * 16, Kp = (K + 63) / 64 * 64;
    *out_elems = Np * Kp;
    return 0;
}

extern "C" int mc_pack_weight_bf16(const void* w, int64_t ldw, void* packed, int N, int K, void* stream) {
    MC_CHECK_ARG(w && packed && N > 0 && K > 0, "mc_pack_weight_bf16: bad arguments N=%d K=%d", N, K);
    const int Np = (N + 15) / 16 * 16, Kp = (K + 63) / 64 * 64;
    const int64_t nfrag = (int64_t)(Np / 16) * (Kp / 32) * 64;
    const int grid = (int)min((int64_t)4096, (nfrag + 255) / 256);
    pack_weight_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)w, (bf16_t*)packed, N, K, Np, Kp, ldw);
    MC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mc_pack_weight_strided_bf16(const void* w, int64_t stride_n, int64_t stride_k, void* packed, int N, int K, void* stream) {
    MC_CHECK_ARG(w && packed && N > 0 && K > 0, "mc_pack_weight_strided_bf16: bad arguments N=%d K=%d", N, K);
    const int Np = (N + 15) / 16 * 16, Kp = (K + 63) / 64 * 64;
    const int64_t nfrag = (int64_t)(Np / 16) * (Kp / 32) * 64;
    const int grid = (int)min((int64_t)4096, (nfrag + 255) / 256);
    pack_weight_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)w, (bf16_t*)packed, N, K, Np, Kp, stride_n, stride_k);
    MC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mc_unpack_weight_bf16(const void* packed, void* w, int N, int K, void* stream) {
    MC_CHECK_ARG(w && packed && N > 0 && K > 0, "mc_unpack_weight_bf16: bad arguments");
    const int Kp = (K + 63) / 64 * 64;
    const int64_t total = (int64_t)N * K;
    const int grid = (int)min((int64_t)4096, (total + 255) / 256);
    unpack_weight_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)packed, (bf16_t*)w, N, K, Kp);
    MC_CHECK_LAUNCH();
    return 0;
}

// ---- per-stream scratch of the tile kernels' rms_out route (one sum of squares per output row and 128-column chunk)
namespace {
struct StreamWs { hipStream_t stream; char* base; };
constexpr size_t kWsFloats = (size_t)12 << 20;               // 48 MiB: M x (N / 128) partial sums (a 134 000-row prefill at N = 4096: 17 MiB)
constexpr size_t kWsBytes = kWsFloats * sizeof(float);
std::vector<StreamWs> g_ws;
std::mutex g_ws_mu;                    // the slot table is touched from every launching thread (tower / pipeline streams, serving threads)
constexpr int kWsPool = 16;            // default, capture, two pipeline and up to four tower streams already make 8
char* g_ws_pool = nullptr;
}  // namespace

// The workspace of a stream: a pool of kWsPool equal workspaces is allocated at the first use (or mc_gemm_reserve_workspace) made while
// the calling stream is not capturing; every stream that needs one is given its own slot then (host bookkeeping only, so a stream met
// for the first time during capture still gets one).  Concurrent streams never share a slot; a stream beyond the pool, or a first use
// inside a capture, gets none (the caller takes its workspace-free route).
static char* stream_workspace(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_ws_mu);
    for (auto& w : g_ws) if (w.stream == s) return w.base;
    if (!g_ws_pool) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
        char* base = nullptr;
        if (hipMalloc((void**)&base, kWsBytes * kWsPool) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        g_ws_pool = base;
    }
    for (int i = 0; i < kWsPool; ++i) {
        char* base = g_ws_pool + kWsBytes * i;
        bool used = false;
        for (auto& w : g_ws) used = used || w.base == base;
        if (!used) { g_ws.push_back({s, base}); return base; }
    }
    return nullptr;
}

// a destroyed stream gives its slot back (its launches have completed: the caller synchronised before destroying it)
extern "C" int mc_gemm_release_workspace(void* stream) {
    std::lock_guard<std::mutex> lock(g_ws_mu);
    for (size_t i = 0; i < g_ws.size(); ++i)
        if (g_ws[i].stream == (hipStream_t)stream) { g_ws.erase(g_ws.begin() + i); break; }
    return 0;
}

extern "C" int mc_gemm_reserve_workspace(void* stream) {
    if (!stream_workspace((hipStream_t)stream)) { mc_set_error("mc_gemm_reserve_workspace: no workspace (allocation failed, first call inside a capture, or more than %d streams)", kWsPool); return 2; }
    return 0;
}

// one launch of the 256x256 kernel over the m-tiles of all groups (M_total = rows over all groups, for the live profile)
static void launch_tile256(const mc_gemm_args* a, const G2Groups& grp, int M_total, const Epilogue& ep, hipStream_t s) {
    const int N = a->N, K = a->K;
    const int ni = tile_ni(grp.tile_start[grp.n], N, a->swiglu != 0 || a->rope != nullptr || ep.ss_parts != nullptr);
    const int tiles_m = grp.tile_start[grp.n], tiles_n = ni == 3 ? (N + 191) / 192 : (N + 255) / 256;
    static bool attr256_set = false;
    const int lds = 2 * G2_STAGE + 1024;            // + 1 KiB nobody reads (destination of the next-tile L2 warm-up)
    if (!attr256_set) {
        (void)hipFuncSetAttribute((const void*)gemm_tile256_kernel<0, 4, 233>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_tile256_kernel<0, 3, 233>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#ifdef MC_PROBES
        (void)hipFuncSetAttribute((const void*)gemm_tile256_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_tile256_kernel<64, 4, 233>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_tile256_kernel<32, 4, 233>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_tile256_kernel<256, 4, 233>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_tile256_kernel<1024, 4, 233>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_tile256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)gemm_tile256_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#endif
        attr256_set = true;
    }
    ProfRec rec{};
    if (g_prof_on) {
        (void)hipEventCreate(&rec.a); (void)hipEventCreate(&rec.b);
        rec.flops = 2.0 * M_total * (double)N * K;
        rec.K = K;
        // algorithmic HBM bytes: x and every group's W read once, the output written once (half as wide with the fused SwiGLU),
        // the residual read once
        rec.bytes = 2.0 * ((double)M_total * K + (double)grp.n * N * K + (double)M_total * (a->swiglu ? N / 2 : N) * (a->out_f32 ? 2 : 1) +
                           (a->residual ? (double)M_total * N : 0.0));
        (void)hipEventRecord(rec.a, s);
    }
    // (probes build: debug word bit 16 forces raster 0, bit 17 forces raster 1)
    int raster = (g_gemm_dbg & 65536) ? 0 : ((g_gemm_dbg & 131072) ? 1 : (g_raster_auto && tiles_m * tiles_n >= g_raster_min_tiles ? 1 : 0));
    // n-slabs (see the kernel): only with the shared-m-group raster and when there is more than one slab
    if (raster == 1 && g_raster_slab > 0 && tiles_n > g_raster_slab && tiles_n > g_raster_slab_min) raster |= g_raster_slab << 8;
    if (ni == 3) {
        gemm_tile256_kernel<0, 3, 233><<<tiles_m * tiles_n, 512, lds, s>>>((const bf16_t*)a->x, a->ldx, grp, N, K, ep, tiles_m, tiles_n, raster);
    } else switch ((g_gemm_dbg >> 3) & 7) {
#ifdef MC_PROBES
#define G2_LAUNCH(A) gemm_tile256_kernel<A><<<tiles_m * tiles_n, 512, lds, s>>>((const bf16_t*)a->x, a->ldx, grp, N, K, ep, tiles_m, tiles_n, raster)
        // debug word bits 3-5: 8 = timing-only ablation without the LDS-DMA (wrong results); 40 = correct results + clock stamps around the main
        // loop (mc_gemm_clock_read); 56 = A/B builds, picked by bits 12-14
        case 1: G2_LAUNCH(1); break;
        case 5: G2_LAUNCH(8); break;
        case 7:                              // A/B builds (results identical): bits 12-14 = 0: the round-1 DMA distribution 2 / 2 / 2 / 2 (with s_setprio)
            if (((g_gemm_dbg >> 12) & 7) == 1) gemm_tile256_kernel<64, 4, 233><<<tiles_m * tiles_n, 512, lds, s>>>((const bf16_t*)a->x, a->ldx, grp, N, K, ep, tiles_m, tiles_n, raster);   // + s_setprio
            else if (((g_gemm_dbg >> 12) & 7) == 2) gemm_tile256_kernel<32, 4, 233><<<tiles_m * tiles_n, 512, lds, s>>>((const bf16_t*)a->x, a->ldx, grp, N, K, ep, tiles_m, tiles_n, raster);   // timing-only: no epilogue
            else if (((g_gemm_dbg >> 12) & 7) == 3) gemm_tile256_kernel<256, 4, 233><<<tiles_m * tiles_n, 512, lds, s>>>((const bf16_t*)a->x, a->ldx, grp, N, K, ep, tiles_m, tiles_n, raster);   // without the next-tile L2 warm-up
            else if (((g_gemm_dbg >> 12) & 7) == 4) gemm_tile256_kernel<1024, 4, 233><<<tiles_m * tiles_n, 512, lds, s>>>((const bf16_t*)a->x, a->ldx, grp, N, K, ep, tiles_m, tiles_n, raster);   // residual in 8-byte loads
            else G2_LAUNCH(64);
            break;
#undef G2_LAUNCH
#endif
        default: gemm_tile256_kernel<0, 4, 233><<<tiles_m * tiles_n, 512, lds, s>>>((const bf16_t*)a->x, a->ldx, grp, N, K, ep, tiles_m, tiles_n, raster); break;
    }
    if (g_prof_on) { (void)hipEventRecord(rec.b, s); g_prof.push_back(rec); }
}

// ---- mc_gemm_args.rms_out: 1/rms of the stored output rows
__global__ __launch_bounds__(256) void rms_parts_kernel(const float* __restrict__ parts, int chunks, int M, int N, float eps, float* __restrict__ rs) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* p = parts + (int64_t)m * chunks;
    float t = 0.f;
    for (int c = 0; c < chunks; ++c) t += p[c];            // column order: deterministic
    rs[m] = rsqrtf(t / (float)N + eps);
}
// The same per-chunk sums from a STORED output (routes other than the 256x256 epilogue), in exactly the epilogue's order - a lane's 32
// values of its column quarter by (half, block pair, element) FMAs, then (q0 + q1) + (q2 + q3) - so that the factor of a row does not
// depend on which kernel produced the row (batch invariance: small launches take other kernels).  4 lanes per (row, chunk).
__global__ __launch_bounds__(256) void rms_chunk_parts_kernel(const bf16_t* __restrict__ out, int64_t ldo, int M, int chunks, float* __restrict__ parts) {
    const int64_t t = blockIdx.x * 256LL + threadIdx.x;
    const int q4 = (int)(t & 3);
    const int64_t rc = t >> 2;
    const bool on = rc < (int64_t)M * chunks;
    const int m = on ? (int)(rc / chunks) : 0, c = on ? (int)(rc % chunks) : 0;
    const bf16_t* p = out + (int64_t)m * ldo + c * 128 + q4 * 4;
    float ss = 0.f;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const bf16x4 lo = *(const bf16x4*)(p + nh * 64 + i * 16), hi = *(const bf16x4*)(p + nh * 64 + i * 16 + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f0 = (float)lo[e], f1 = (float)hi[e];
                ss = fmaf(f0, f0, ss);
                ss = fmaf(f1, f1, ss);
            }
        }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    if (on && q4 == 0) parts[(int64_t)m * chunks + c] = ss;
}
static bool rms_out_args_ok(const mc_gemm_args* a) { return !a->out_f32 && !a->swiglu && !a->rope && a->split_k <= 1 && a->rms_out_eps > 0.f; }
// the epilogue route needs whole 256-column tiles of the 256x256 kernel, no activation, and the partial sums to fit the stream's workspace
static float* rms_parts_buffer(const mc_gemm_args* a, int64_t M_total, hipStream_t s) {
    if (a->N % 128 || a->ldo % 4 || ((uintptr_t)a->out % 8) || (size_t)M_total * (a->N / 128) > kWsFloats) return nullptr;
    return (float*)stream_workspace(s);
}
static float* rms_out_parts(const mc_gemm_args* a, int64_t M_total, hipStream_t s) {
    if (!a->rms_out || a->N % 256 || a->act != MC_ACT_NONE || a->ldo % 8 || ((uintptr_t)a->out % 16) || (g_gemm_dbg & (1 << 29))) return nullptr;
    return rms_parts_buffer(a, M_total, s);
}
// parts: what the epilogue left (null: another route ran).  Without a workspace (first launch inside a capture) or for N that is not a
// multiple of 128 the plain row pass computes the factor (other fp32 summation order).
static int rms_out_after(const mc_gemm_args* a, int64_t row0, int M, float* parts, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int chunks = a->N / 128;
    if (!parts && (parts = rms_parts_buffer(a, M, s)) != nullptr) {
        const int64_t threads = (int64_t)M * chunks * 4;
        rms_chunk_parts_kernel<<<(int)((threads + 255) / 256), 256, 0, s>>>((const bf16_t*)a->out + row0 * a->ldo, a->ldo, M, chunks, parts);
    }
    if (parts) {
        rms_parts_kernel<<<(M + 255) / 256, 256, 0, s>>>(parts, chunks, M, a->N, a->rms_out_eps, a->rms_out + row0);
        MC_CHECK_LAUNCH();
        return 0;
    }
    return mc_rms_scale_bf16((const char*)a->out + row0 * a->ldo * 2, a->ldo, a->rms_out + row0, M, a->N, a->rms_out_eps, stream);
}

// ---- RoPE + scatter epilogue of the q|k|v projection (mc_gemm_args.rope)
static bool rope_args_ok(const mc_gemm_args* a) {
    const mc_rope_scatter* r = a->rope;
    return r->row_b && r->row_pos && r->row_t && r->cos_table && r->sin_table && r->q_out && r->k_cache && r->v_cache && r->H > 0 && r->Hkv > 0 &&
           r->D % 16 == 0 && a->N == (r->H + 2 * r->Hkv) * r->D && !a->out_f32 && !a->swiglu && !a->residual && a->act == MC_ACT_NONE &&
           a->split_k <= 1 && a->ldo % 8 == 0;
}
// the register route: a wave's 128 columns of a 256-column tile must be one head.  debug word bit 31 keeps the separate launch (A/B)
static bool rope_in_epilogue(const mc_gemm_args* a) {
    const mc_rope_scatter* r = a->rope;
    return r && r->D == 128 && a->N % 256 == 0 && !(g_gemm_dbg & (1u << 31));
}
static void rope_fill(Epilogue& ep, const mc_rope_scatter* r, int64_t row0) {
    ep.rope = Epilogue::Rope{r->row_b + row0, r->row_pos + row0, r->row_t + row0, r->cos_table, r->sin_table, (bf16_t*)r->q_out,
                             (bf16_t*)r->k_cache, (bf16_t*)r->v_cache, r->H, r->Hkv, r->Lq, r->Smax};
}
// the separate route over rows [row0, row0 + M) of the stored projection
static int rope_after(const mc_gemm_args* a, int64_t row0, int M, void* stream) {
    const mc_rope_scatter* r = a->rope;
    return mc_rope_kv_bf16((const char*)a->out + row0 * a->ldo * 2, a->ldo, r->row_b + row0, r->row_pos + row0, r->row_t + row0, r->cos_table,
                           r->sin_table, r->q_out, r->k_cache, r->v_cache, M, r->H, r->Hkv, r->D, r->Lq, r->Smax, stream);
}

extern "C" int mc_gemm_ex_bf16(const mc_gemm_args* a, void* stream) {
    MC_CHECK_ARG(a, "mc_gemm_ex_bf16: null argument block");
    const void* x = a->x; const int64_t ldx = a->ldx; const void* w_packed = a->w_packed;
    const int M = a->M, N = a->N, K = a->K;
    MC_CHECK_ARG(x && w_packed && a->out, "mc_gemm_bf16: null pointer");
    MC_CHECK_ARG(M > 0 && N > 0 && K > 0, "mc_gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    MC_CHECK_ARG(K % 64 == 0, "mc_gemm_bf16: K=%d must be a multiple of 64 (pad activations/weights)", K);
    MC_CHECK_ARG(N % 4 == 0, "mc_gemm_bf16: N=%d must be a multiple of 4", N);
    MC_CHECK_ARG(ldx % 8 == 0 && ((uintptr_t)x % 16) == 0, "mc_gemm_bf16: x must be 16-byte aligned rows (ldx=%lld)", (long long)ldx);
    MC_CHECK_ARG(a->ldo % 4 == 0, "mc_gemm_bf16: ldo=%lld must be a multiple of 4", (long long)a->ldo);
    MC_CHECK_ARG(!a->residual || a->ldr % 4 == 0, "mc_gemm_bf16: ldr must be a multiple of 4");
    MC_CHECK_ARG(!a->swiglu || (N % 32 == 0 && !a->bias && !a->residual && !a->out_f32 && a->act == MC_ACT_NONE),
                 "mc_gemm_ex_bf16: swiglu needs N %% 32 == 0 (gate/up interleaved per 16 rows) and a plain bf16 output");
    MC_CHECK_ARG(a->split_k <= 1, "mc_gemm_ex_bf16: split_k must be 1 (or < 0: automatic, tile family)");
    MC_CHECK_ARG(a->family >= MC_GEMM_AUTO && a->family <= MC_GEMM_TILE, "mc_gemm_ex_bf16: family %d", a->family);
    const bool strip = a->family == MC_GEMM_STRIP || (a->family == MC_GEMM_AUTO && M <= 64 && a->split_k >= 0);
    MC_CHECK_ARG(!(a->rms_eps > 0.f) || (strip && !a->row_scale),
                 "mc_gemm_ex_bf16: rms_eps (in-kernel RMS factor) is the strip family's (M <= 64, or family = MC_GEMM_STRIP), without row_scale");
    MC_CHECK_ARG(!a->rope || rope_args_ok(a), "mc_gemm_ex_bf16: rope needs N = (H + 2 Hkv) D, a plain bf16 output (ldo %% 8 == 0) and every pointer");
    Epilogue ep{(const bf16_t*)a->bias, (const bf16_t*)a->residual, a->ldr, a->out, a->ldo, a->act, a->out_f32, a->alpha, a->beta,
                a->row_scale, a->swiglu, a->rms_eps > 0.f ? a->rms_eps : 0.f};
    MC_CHECK_ARG(!a->rms_out || rms_out_args_ok(a), "mc_gemm_ex_bf16: rms_out needs a bf16 output without SwiGLU / rope / split_k and rms_out_eps > 0");
    hipStream_t s = (hipStream_t)stream;
    bool rope_pending = a->rope != nullptr;
    float* ss_parts = nullptr;
    if (strip) {
        // the strip family (gemm_strip.hip); above 64 rows as slices of 64 rows: the weights are streamed once per slice (decode batches of
        // that size are attention-bound) and every row sees exactly the launch it would see in a smaller batch
        for (int r0 = 0; r0 < M; r0 += 64) {
            Epilogue e1 = ep;
            e1.out = (char*)a->out + (int64_t)r0 * a->ldo * (a->out_f32 ? 4 : 2);
            if (a->residual) e1.residual = (const bf16_t*)a->residual + (int64_t)r0 * a->ldr;
            if (a->row_scale) e1.row_scale = a->row_scale + r0;
            const int rc = mc_strip_launch((const bf16_t*)x + (int64_t)r0 * ldx, ldx, (const bf16_t*)w_packed, min(64, M - r0), N, K, e1, s);
            if (rc) return rc;
        }
    } else if (use_tile256(M, N, K)) {
        G2Groups grp{};
        grp.n = 1; grp.tile_start[0] = 0; grp.tile_start[1] = (M + 255) / 256; grp.row_start[0] = 0; grp.row_start[1] = M;
        grp.wp[0] = (const bf16_t*)w_packed;
        if (rope_in_epilogue(a)) { rope_fill(ep, a->rope, 0); rope_pending = false; }
        if ((ss_parts = rms_out_parts(a, M, s)) != nullptr) { ep.ss_parts = ss_parts; ep.ss_chunks = N / 128; }
        launch_tile256(a, grp, M, ep, s);
    } else {
        const int tiles_m = (M + TM - 1) / TM, tiles_n = (N + TN - 1) / TN;
        static bool attr_set = false;
        const int lds = 4 * TILE_BYTES;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)gemm_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_set = true;
        }
        // split_k < 0 = "auto": under-filled grids with a long reduction (the rank projections x A^T and dy B of the LoRA branches in the
        // finetune step: N = n_adapters * r columns, K = 4096 ... 11008) are split along K so that ~2 workgroups per CU are resident; slabs
        // are summed in fixed order (deterministic).  Opt-in because the split depends on M: inference keeps batch-invariant rounding.
        const int tiles = tiles_m * tiles_n, nt = K / TK;
        int S = 1;
        if (a->split_k < 0 && !a->swiglu && !(g_gemm_dbg & (2 | 512)) && tiles < 192 && nt >= 16) {
            S = min(min(8, 512 / tiles), nt / 8);
            if (S < 2) S = 1;
        }
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (S > 1 && (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) S = 1;     // no allocation while capturing
        float* slabs = S > 1 ? splitk_workspace((size_t)S * M * N) : nullptr;
        if (S > 1 && slabs) {
            Epilogue raw{nullptr, nullptr, 0, slabs, N, MC_ACT_NONE, 1, 1.0f, 0.0f, nullptr, 0, 0.f};
            gemm_tile_kernel<<<dim3(tiles, S), 256, lds, s>>>((const bf16_t*)x, ldx, (const bf16_t*)w_packed, M, N, K, raw, tiles_m, tiles_n, g_gemm_dbg);
            const int64_t total = (int64_t)M * (N >> 2);
            splitk_reduce_kernel<<<(int)min((int64_t)2048, (total + 255) / 256), 256, 0, s>>>(slabs, S, M, N, ep);
        } else {
            gemm_tile_kernel<<<tiles, 256, lds, s>>>((const bf16_t*)x, ldx, (const bf16_t*)w_packed, M, N, K, ep, tiles_m, tiles_n, g_gemm_dbg);
        }
    }
    MC_CHECK_LAUNCH();
    if (rope_pending) return rope_after(a, 0, M, stream);
    if (a->rms_out) return rms_out_after(a, 0, M, ss_parts, stream);
    return 0;
}

extern "C" int mc_gemm_bf16(const void* x, int64_t ldx, const void* w_packed, const void* bias, const void* residual,
                            int64_t ldr, void* out, int64_t ldo, int M, int N, int K, int act, int out_f32, float alpha,
                            float beta, void* stream) {
    mc_gemm_args a;
    a.x = x; a.ldx = ldx; a.w_packed = w_packed; a.bias = bias; a.residual = residual; a.ldr = ldr; a.out = out; a.ldo = ldo;
    a.M = M; a.N = N; a.K = K; a.act = act; a.out_f32 = out_f32; a.alpha = alpha; a.beta = beta;
    a.row_scale = nullptr; a.swiglu = 0; a.split_k = 1; a.rms_eps = 0.f; a.rope = nullptr; a.rms_out = nullptr; a.rms_out_eps = 0.f;
    a.family = MC_GEMM_AUTO;
    return mc_gemm_ex_bf16(&a, stream);
}

// Routed LocalLoRA linear over adapter-grouped rows: rows [group_start[g], group_start[g+1]) of x / out / residual / row_scale use
// w_packed[g] (group_start: n_groups+1 host ints, w_packed: n_groups host pointers).  Large problems run as ONE launch of the 256x256
// kernel (each group padded to whole m-tiles), so small groups no longer pay their own launch and partial last round; otherwise every
// group is a separate mc_gemm_ex_bf16 call.  args->w_packed / args->M are ignored.
extern "C" int mc_gemm_grouped_bf16(const mc_gemm_args* args, int n_groups, const int32_t* group_start, const void* const* w_packed,
                                    void* stream) {
    MC_CHECK_ARG(args && group_start && w_packed && n_groups >= 1, "mc_gemm_grouped_bf16: bad arguments");
    const int M_total = group_start[n_groups] - group_start[0];
    int64_t tiles = 0;
    int ng = 0;
    for (int g = 0; g < n_groups; ++g) {
        const int mg = group_start[g + 1] - group_start[g];
        MC_CHECK_ARG(mg >= 0 && w_packed[g], "mc_gemm_grouped_bf16: bad group %d", g);
        if (mg > 0) { tiles += (mg + 255) / 256; ++ng; }
    }
    const int N = args->N, K = args->K;
    const bool one_launch = args->family != MC_GEMM_STRIP && ng >= 1 && ng <= 8 && M_total > 64 && K >= 128 && K % 64 == 0 && !(g_gemm_dbg & 2) && args->split_k <= 1 &&
                            ((g_gemm_dbg & 4) || tiles * ((N + 255) / 256) >= 144);
    if (!one_launch) {
        for (int g = 0; g < n_groups; ++g) {
            const int r0 = group_start[g], mg = group_start[g + 1] - r0;
            if (mg <= 0) continue;
            mc_gemm_args a = *args;
            const int esz_o = a.out_f32 ? 4 : 2;
            a.x = (const char*)args->x + (int64_t)r0 * args->ldx * 2;
            a.out = (char*)args->out + (int64_t)r0 * args->ldo * esz_o;
            if (args->residual) a.residual = (const char*)args->residual + (int64_t)r0 * args->ldr * 2;
            if (args->row_scale) a.row_scale = args->row_scale + r0;
            a.w_packed = w_packed[g]; a.M = mg;
            if (args->rms_out) a.rms_out = args->rms_out + r0;
            mc_rope_scatter rg;
            if (args->rope) {
                rg = *args->rope;
                rg.row_b += r0; rg.row_pos += r0; rg.row_t += r0;
                a.rope = &rg;
            }
            const int rc = mc_gemm_ex_bf16(&a, stream);
            if (rc) return rc;
        }
        return 0;
    }
    MC_CHECK_ARG(args->x && args->out && N > 0 && N % 4 == 0, "mc_gemm_grouped_bf16: bad arguments");
    MC_CHECK_ARG(args->ldx % 8 == 0 && ((uintptr_t)args->x % 16) == 0 && args->ldo % 4 == 0 && (!args->residual || args->ldr % 4 == 0),
                 "mc_gemm_grouped_bf16: alignment");
    MC_CHECK_ARG(!args->swiglu || (N % 32 == 0 && !args->bias && !args->residual && !args->out_f32 && args->act == MC_ACT_NONE),
                 "mc_gemm_grouped_bf16: swiglu needs N %% 32 == 0 and a plain bf16 output");
    // the kernel indexes rows from the base pointers: shift them to row group_start[0] once
    const int base = group_start[0];
    mc_gemm_args a = *args;
    a.x = (const char*)args->x + (int64_t)base * args->ldx * 2;
    a.out = (char*)args->out + (int64_t)base * args->ldo * (args->out_f32 ? 4 : 2);
    if (args->residual) a.residual = (const char*)args->residual + (int64_t)base * args->ldr * 2;
    if (args->row_scale) a.row_scale = args->row_scale + base;
    MC_CHECK_ARG(!args->rope || rope_args_ok(args), "mc_gemm_grouped_bf16: rope needs N = (H + 2 Hkv) D, a plain bf16 output (ldo %% 8 == 0) and every pointer");
    Epilogue ep{(const bf16_t*)a.bias, (const bf16_t*)a.residual, a.ldr, a.out, a.ldo, a.act, a.out_f32, a.alpha, a.beta, a.row_scale, a.swiglu, 0.f};
    const bool rope_fused = rope_in_epilogue(args);
    if (rope_fused) rope_fill(ep, args->rope, base);
    MC_CHECK_ARG(!args->rms_out || rms_out_args_ok(args), "mc_gemm_grouped_bf16: rms_out needs a bf16 output without SwiGLU / rope / split_k and rms_out_eps > 0");
    float* ss_parts = rms_out_parts(&a, M_total, (hipStream_t)stream);
    if (ss_parts) { ep.ss_parts = ss_parts; ep.ss_chunks = N / 128; }
    G2Groups grp{};
    int t = 0, k = 0;
    for (int g = 0; g < n_groups; ++g) {
        const int mg = group_start[g + 1] - group_start[g];
        if (mg <= 0) continue;
        grp.tile_start[k] = t; grp.row_start[k] = group_start[g] - base; grp.wp[k] = (const bf16_t*)w_packed[g];
        t += (mg + 255) / 256;
        ++k;
        grp.row_start[k] = group_start[g + 1] - base;     // end of this group (= start of the next non-empty one: groups are contiguous)
    }
    grp.n = k; grp.tile_start[k] = t;
    launch_tile256(&a, grp, M_total, ep, (hipStream_t)stream);
    MC_CHECK_LAUNCH();
    if (args->rope && !rope_fused) return rope_after(args, base, M_total, stream);
    if (args->rms_out) return rms_out_after(args, base, M_total, ss_parts, stream);
    return 0;
}
