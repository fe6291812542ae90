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
#include <stdlib.h>
#include <string.h>

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
#if MC_STORAGE_IS_F16
    // IEEE half: the two neighbours of v on the half grid are its nearest value and the next one on the other side of v; v is rounded to the
    // farther one with probability = its distance from the nearer one over the spacing (unbiased: E = v)
    const bf16_t h0 = (bf16_t)v;
    const float f0 = (float)h0;
    if (f0 == v || !(fabsf(v) < 65504.0f)) return h0;                            // on the grid, or nothing finite beyond it
    uint16_t b0 = __builtin_bit_cast(uint16_t, h0);
    const bool away = fabsf(f0) < fabsf(v);                                       // v lies beyond h0 (away from zero)
    uint16_t b1;
    if ((b0 & 0x7fffu) == 0) b1 = (uint16_t)((v < 0.f ? 0x8000u : 0u) | 1u);      // from (signed) zero: the smallest subnormal towards v
    else b1 = away ? (uint16_t)(b0 + 1) : (uint16_t)(b0 - 1);
    const bf16_t h1 = __builtin_bit_cast(bf16_t, b1);
    const float f1 = (float)h1;
    const float frac = (v - f0) / (f1 - f0);                                      // in (0, 0.5]: how far v sits from its nearest neighbour
    const uint32_t r = compose_hash(seed ^ compose_hash((uint32_t)n * 0x9E3779B1U + (uint32_t)k)) & 0xFFFFU;
    return ((float)r + 0.5f) * (1.0f / 65536.0f) < frac ? h1 : h0;
#else
    uint32_t bits = __builtin_bit_cast(uint32_t, v);
    if ((bits & 0x7f800000U) == 0x7f800000U || (bits & 0x7fff0000U) == 0x7f7f0000U) return (bf16_t)v;   // inf / nan, and the largest finite bf16 (the carry would make it inf): as the plain cast
    const uint32_t r = compose_hash(seed ^ compose_hash((uint32_t)n * 0x9E3779B1U + (uint32_t)k)) & 0xFFFFU;
    bits += r;                                                                // sign-magnitude: the magnitude goes up with probability frac / 2^16
    const uint16_t hi = (uint16_t)(bits >> 16);
    return __builtin_bit_cast(bf16_t, hi);
#endif
}

// Round 5: ONE pass per linear for ALL routed adapters.  The round-4 kernel ran once per adapter (4 launches per linear for the 3-way composed
// model: 4 x (read W + write W')), read W in 8-byte pieces of 16 different rows per wave instruction and re-read it element by element for
// the retention statistic: 0.8 TB/s.  Here a workgroup reads its W tile ONCE, with 16-byte loads, keeps it in registers, and writes every
// adapter's W' from it - 1 read + n_out writes instead of n_out x (read + write) - each output as whole 1-KiB blocks of the packed layout
// (16 bytes per lane: v_permlane16_swap pairs the two 16-column blocks of a lane pair, the exchange the GEMM epilogue uses).
// workgroup = 4 waves; tile = 16 NR rows (n) x 256 cols (k); wave w owns cols [64w, 64w + 64)
#define MC_MAX_OUTS 6

struct ComposeMultiParams {
    const bf16_t* w; int64_t ldw;            // may be null (-> pure delta)
    const bf16_t* at[MC_MAX_TERMS];          // [K, r]
    const bf16_t* bm[MC_MAX_TERMS];          // [N, r]
    float scale[MC_MAX_TERMS];
    int n_terms, r, n_out;
    bf16_t* out_packed[MC_MAX_OUTS];         // [ceil16(N)/16][Kp/32][64][8] each
    bf16_t* out_rowmajor[MC_MAX_OUTS];       // optional row-major copies (ld = ldo)
    uint32_t term_mask[MC_MAX_OUTS];         // bit m: term m belongs to this output (summed in term order)
    uint32_t dither_seed[MC_MAX_OUTS];       // 0: round to nearest even; else unbiased rounding (compose_round)
    float* retention[MC_MAX_OUTS];           // optional per-output [gridDim.y][gridDim.x][2] partial sums (see mc_hip.h)
    int64_t ldo;
    int N, K, Kp;
    const float* col_scale;                  // [K] fp32 or null
    int nb_stride, nb_offset;
};

// FAST: K % 8 == 0 and 16-byte aligned rows of W / the row-major copies - every 8-element piece is loaded and stored whole, no per-element
// guards (the shapes of the LLM); the general instantiation guards every element.
// Loop order (register budget: 2 waves per SIMD): per output, per term, the term's A^T fragments of the wave's 64 columns are loaded ONCE
// (RS x 4 fragments) and reused against the NR row blocks, whose B fragments stream through; a multi-term output keeps its running fp32
// total in LDS (16 bytes per lane and 16 x 16 block, conflict-free), a single-term output never leaves the registers.
// WT: the outputs' 16-byte stores written THROUGH (sc1: the line is not kept in the XCD's L2).  A launch writes n_out times the bytes it reads;
// kept in the write-back L2 that stream evicts the LoRA factors every tile re-reads (12 x 1 MiB against a 4-MiB L2), which then come from the
// Infinity Cache over the fabric - measured with plain stores: 0.9 TB/s, the fabric-side re-reads 3-5 x the algorithmic bytes.
__device__ __forceinline__ void compose_store16(void* p, u32x4 v, bool wt) {
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else *(u32x4*)p = v;
}

template <int NR, bool FAST, bool WT>
__global__ __launch_bounds__(256, 2) void compose_multi_kernel(ComposeMultiParams p_by_value) {
    // The argument block is read where it lies, in the kernarg segment: its arrays are indexed by the run-time output / term number, and a
    // by-value aggregate indexed dynamically is first loaded whole into ~100 SGPRs (130 of them then spilled into vector registers, which in
    // turn spilled to scratch); from memory each use is one scalar load with a computed offset.
    (void)p_by_value;
    const ComposeMultiParams& p = *(const ComposeMultiParams*)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int RS = 4;                                  // k-steps of 32 along the rank held in registers at a time (r = 128: all of them)
    __shared__ __attribute__((aligned(16))) f32x4 tots[4][NR][4][64];
    __shared__ float red[MC_MAX_OUTS][4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * (16 * NR);
    const int k0 = blockIdx.x * 256 + wave * 64;
    const bool wave_on = k0 < p.Kp;
    const int kblocks = p.Kp >> 5;
    const int np16 = (p.N + 15) & ~15;
    // this lane's 16-byte piece of a block pair tp (blocks 2 tp, 2 tp + 1 of the wave's four 16-column blocks): 8 consecutive k of row n
    const int k8off = (g & 1) * 16 + (g >> 1) * 8;
    float cs[4][4];
    if (wave_on) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = k0 + t * 16 + g * 4;
            if (FAST) {
                const f32x4 c4 = (p.col_scale && k < p.K) ? *(const f32x4*)(p.col_scale + k) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) cs[t][j] = c4[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) cs[t][j] = p.col_scale ? p.col_scale[min(k + j, p.K - 1)] : 1.0f;
            }
        }
    }
    // W of row block i in the accumulator layout (lane: n = c16, k = 16 t + 4 g .. + 3): 16-byte loads of the 8 columns the lane will STORE,
    // swapped back (the exchange is its own inverse).  The first output's loads come from HBM, the later ones find the tile in the L2 /
    // Infinity Cache: W crosses the HBM interface once per linear.
    auto load_w = [&](int i, bf16x4 (&wv)[4]) {
        const int n = n0 + i * 16 + c16;
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            const int k = k0 + tp * 32 + k8off;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (p.w && n < p.N) {
                if (FAST) {
                    if (k < p.K) v = *(const u32x4*)(p.w + (int64_t)n * p.ldw + k);
                } else {
                    bf16x8 e;
#pragma unroll
                    for (int j = 0; j < 8; ++j) e[j] = (k + j < p.K) ? p.w[(int64_t)n * p.ldw + k + j] : (bf16_t)0.0f;
                    v = __builtin_bit_cast(u32x4, e);
                }
            }
            auto r0 = __builtin_amdgcn_permlane16_swap(v[0], v[2], false, false);
            auto r1 = __builtin_amdgcn_permlane16_swap(v[1], v[3], false, false);
            const u32x2 lo = {r0[0], r1[0]}, hi = {r0[1], r1[1]};
            wv[2 * tp] = __builtin_bit_cast(bf16x4, lo);
            wv[2 * tp + 1] = __builtin_bit_cast(bf16x4, hi);
        }
    };
    for (int a = 0; a < p.n_out; ++a) {
        float ret_num = 0.f, ret_den = 0.f;
        if (wave_on) {
            const uint32_t mask = p.term_mask[a];
            const uint32_t seed = p.dither_seed[a];
            const bool want_ret = p.retention[a] != nullptr && p.w != nullptr;
            bf16_t* const outp = p.out_packed[a];
            bf16_t* const outr = p.out_rowmajor[a];
            const int last_m = mask ? 31 - __builtin_clz(mask) : -1;
            const int first_m = mask ? __builtin_ctz(mask) : -1;
            // the epilogue of row block i: total (fp32) + W, column scale, ONE rounding, store
            auto finish = [&](int i, const f32x4 (&tot)[4], const bf16x4 (&wv)[4]) {
                const int n = n0 + i * 16 + c16;
                const int nb = n >> 4;                           // wave-uniform (c16 < 16)
                if (nb * 16 >= np16) return;
                const bool inb = n < p.N;
#pragma unroll
                for (int tp = 0; tp < 2; ++tp) {
                    bf16x4 o[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int t = 2 * tp + h;
                        const int k = k0 + t * 16 + g * 4;
                        float r4[4] = {tot[t][0], tot[t][1], tot[t][2], tot[t][3]};
                        if (p.w && inb) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) r4[j] += (float)wv[t][j];
                        }
                        if (p.col_scale) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) r4[j] *= cs[t][j];
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[h][j] = (inb && k + j < p.K) ? compose_round(r4[j], seed, n, k + j) : (bf16_t)0.0f;
                        if (want_ret && inb) {
                            // the composed weight against the base weight rounded the same way: the part of (W' - bf16(W c)) that lies along dW c
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (k + j < p.K) {
                                    const float d = tot[t][j] * cs[t][j];
                                    const float moved = (float)o[h][j] - (float)(bf16_t)((float)wv[t][j] * cs[t][j]);
                                    ret_num = fmaf(moved, d, ret_num);
                                    ret_den = fmaf(d, d, ret_den);
                                }
                        }
                    }
                    // 16 bytes per lane: even g the 8 columns (g >> 1) * 8 .. of block 2 tp, odd g of block 2 tp + 1
                    const u32x2 pa = __builtin_bit_cast(u32x2, o[0]), pb = __builtin_bit_cast(u32x2, o[1]);
                    auto r0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
                    auto r1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
                    const u32x4 ov = {r0[0], r1[0], r0[1], r1[1]};
                    const int k8 = k0 + tp * 32 + k8off;
                    const int kb = k8 >> 5, q = (k8 & 31) >> 3;
                    bf16_t* dst = outp + ((int64_t)(nb * p.nb_stride + p.nb_offset) * kblocks + kb) * 512 + ((q << 4) | (n & 15)) * 8;
                    compose_store16(dst, ov, WT);
                    if (outr && inb) {
                        if (FAST) {
                            if (k8 < p.K) *(u32x4*)(outr + (int64_t)n * p.ldo + k8) = ov;
                        } else {
                            const bf16x8 e = __builtin_bit_cast(bf16x8, ov);
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (k8 + j < p.K) outr[(int64_t)n * p.ldo + k8 + j] = e[j];
                        }
                    }
                }
            };
            if (mask == 0) {                                     // no LoRA term: W' = bf16(W c)
                const f32x4 zero[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
                for (int i = 0; i < NR; ++i) {
                    bf16x4 wv0[4];
                    load_w(i, wv0);
                    finish(i, zero, wv0);
                }
            }
            // One term pass = every load it needs issued back to back (the term's A^T fragments, the B fragments of ALL row blocks and, on an
            // output's last term, the W pieces the epilogue adds), then the MFMA chains: one exposed memory latency per pass instead of one per
            // row block (round 5, first version: 0.85 TB/s - each wave waited ~2 us for four 16-byte loads sixteen times per tile).
            // Ranks beyond 32 RS are processed in chunks of 32 RS, the chunk sums added in fp32 (r <= 128: one chunk, one MFMA chain).
            const int n_chunks = (p.r + 32 * RS - 1) / (32 * RS);
            for (int m = 0; m < p.n_terms; ++m) {
                if (!((mask >> m) & 1u)) continue;
                const bf16_t* at = p.at[m];
                const bf16_t* bm = p.bm[m];
                const float s = p.scale[m];
                for (int ch = 0; ch < n_chunks; ++ch) {
                    const int rs0 = ch * 32 * RS;
                    const bool first = m == first_m && ch == 0, last = m == last_m && ch == n_chunks - 1;
                    bf16x8 af[RS][4];
#pragma unroll
                    for (int q = 0; q < RS; ++q)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int k = min(k0 + t * 16 + c16, p.K - 1);
                            af[q][t] = *(const bf16x8*)(at + (int64_t)k * p.r + min(rs0 + q * 32, p.r - 32) + g * 8);      // (steps beyond r are skipped below)
                        }
                    // row blocks in groups of NG: the group's B fragments and W pieces are requested together (a real loop over the groups keeps
                    // the register count at two waves per SIMD; fully unrolled, hipcc spilled 186 registers)
                    constexpr int NG = 2;
#pragma unroll 1
                    for (int ig = 0; ig < NR; ig += NG) {
                        bf16x8 bf[NG][RS];
                        bf16x4 wv[NG][4];
#pragma unroll
                        for (int j = 0; j < NG; ++j) {
                            const int n = min(n0 + (ig + j) * 16 + c16, p.N - 1);
#pragma unroll
                            for (int q = 0; q < RS; ++q) bf[j][q] = *(const bf16x8*)(bm + (int64_t)n * p.r + min(rs0 + q * 32, p.r - 32) + g * 8);
                        }
                        if (last) {
#pragma unroll
                            for (int j = 0; j < NG; ++j) load_w(ig + j, wv[j]);
                        }
#pragma unroll
                        for (int j = 0; j < NG; ++j) {
                            const int i = ig + j;
                            f32x4 acc[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int q = 0; q < RS; ++q) {
                                if (rs0 + q * 32 < p.r) {
#pragma unroll
                                    for (int t = 0; t < 4; ++t) acc[t] = mc_mfma_16x16x32(af[q][t], bf[j][q], acc[t]);
                                }
                            }
                            f32x4 tot[4];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                tot[t] = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : tots[wave][i][t][lane];
                                tot[t] += acc[t] * s;
                            }
                            if (last) finish(i, tot, wv[j]);
                            else {
#pragma unroll
                                for (int t = 0; t < 4; ++t) tots[wave][i][t][lane] = tot[t];
                            }
                        }
                    }
                }
            }
        }
        if (p.retention[a]) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                ret_num += __shfl_xor(ret_num, o, 64);
                ret_den += __shfl_xor(ret_den, o, 64);
            }
            if (lane == 0) { red[a][wave][0] = ret_num; red[a][wave][1] = ret_den; }
        }
    }
    __syncthreads();
    if (threadIdx.x < p.n_out && p.retention[threadIdx.x]) {
        const int a = threadIdx.x;
        float* dst = p.retention[a] + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        dst[0] = (red[a][0][0] + red[a][1][0]) + (red[a][2][0] + red[a][3][0]);            // fixed order: reproducible
        dst[1] = (red[a][0][1] + red[a][1][1]) + (red[a][2][1] + red[a][3][1]);
    }
}

// ------------------------------------------------------------------------------------------
// Round 6: the composition as ONE launch over MANY linears (descriptor table) with the LoRA factors staged through LDS.
// ------------------------------------------------------------------------------------------
// What bounded compose_multi_kernel at 0.12 of HBM (profiles/r05_probes/compose_probe.json): a 64 x 256 tile per workgroup re-reads its
// A^T (64 KiB) and B (16 KiB) fragments per term straight from L2 - 480 KiB of factor traffic against 160 KiB of W + W' per tile at 6 terms -
// every wave waits for its own fragments before each MFMA group, W is re-read per output, and a model is 224 launches of 180-530 us.
// Here: tile 128 (n) x 128 (k), 8 waves (wave (wn, wk): rows 64 wn .., columns 32 wk ..); per (output, term) the factors of up to 128
// ranks go in stages of 64 ranks - A^T 128 x 64 + B 128 x 64 = 32 KiB - through a 4-stage LDS ring by LDS-DMA in whole 128-byte lines (the
// 16-byte piece p of row r of an 8-row octet at slot 8 r + (p ^ 2 (r >> 1)): conflict-free for the DMA's lane order and for the fragment
// reads) and are shared by the waves of a row / column; W is loaded ONCE per tile into registers and every output is written from it; three
// stages of DMA are in flight under the MFMAs of one, one barrier per stage.  blockIdx.x -> (linear, tile) through the descriptor table: one
// launch per model.  The arithmetic per element is the kernel's above - one v_mfma_f32_16x16x32 chain over each chunk of 128 ranks from
// zero, total += chain * scale in term order, + W, x column scale, one rounding - so the weights are BIT-identical.
// Needs K % 64 == 0 (no padded columns), r % 64 == 0, 16-byte aligned rows of W / A^T / B; everything else takes the kernel above.
constexpr int CT_N = 128, CT_K = 128;
constexpr int CT_HALF_BYTES = (CT_K + CT_N) * 64 * 2;             // one stage = 64 ranks of a term: A^T octets [16] | B octets [16], 1 KiB each (32 KiB)
constexpr int CT_RING = 4;                                        // stages in the LDS ring: three in flight under the MFMAs of one (128 KiB)
constexpr int CT_MAX_STAGES = 96;                                 // (output, term) pairs x 64-rank halves of one linear (6 outputs, 8 terms, r <= 256 in practice)

// What a stage needs, in ONE 32-byte record (the stage loop reads it with two LDS loads; indexing the argument block's tables instead -
// pair -> term -> pointers, output -> mask / pointers - was 8-10 DEPENDENT LDS round trips per stage and per epilogue call: 20 us of a tile)
struct CtStage { const bf16_t* at; const bf16_t* bm; float scale; int flags; int64_t pad; };      // at / bm: + the stage's rank offset
enum { CT_CHUNK_END = 1, CT_FIRST = 2, CT_LAST = 4 };                                               // flags; bits 8..: the output index
struct CtOut { bf16_t* outp; bf16_t* outr; float* ret; uint32_t seed, mask; };
struct ComposeDesc {
    const bf16_t* w; const float* col_scale; int64_t ldw, ldo;
    int N, K, r, n_out, nb_stride, nb_offset;
    int tile_start, tiles_n, tiles_k, n_stages;
    int pad_[2];                                                  // sizeof % 16 == 0: copied to LDS in 16-byte pieces
    CtOut out[MC_MAX_OUTS];
    CtStage st[CT_MAX_STAGES];
};

typedef __attribute__((address_space(3))) void ct_lds_void;
typedef const __attribute__((address_space(1))) void ct_gbl_void;

// wave-uniform values read from LDS arrive in vector registers: move them to the scalar file (they are indices, bounds and base pointers)
__device__ __forceinline__ int ct_u(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t ct_u64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <class T> __device__ __forceinline__ T* ct_up(T* q) { return (T*)ct_u64((int64_t)q); }

template <int N_> __device__ __forceinline__ void ct_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <bool WT>
__global__ __launch_bounds__(512, 2) void compose_tile_kernel(const ComposeDesc* __restrict__ descs, int n_desc, int total_tiles) {
    extern __shared__ __attribute__((aligned(1024))) char ct_smem[];          // CT_RING stages of CT_HALF_BYTES
    __shared__ float red[MC_MAX_OUTS][8][2];
    __shared__ __attribute__((aligned(16))) ComposeDesc sdesc;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int wn = wave >> 2, wk = wave & 3;
    const int k8off = (g & 1) * 16 + (g >> 1) * 8;
    const int row8 = lane >> 3, piece = (lane & 7) ^ (2 * (row8 >> 1));
    // fragment read slots (16-byte units inside a stage): row c16 of the block's octet pair, piece 4 q + g
    int fslot[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) fslot[q] = (c16 >> 3) * 64 + (c16 & 7) * 8 + ((4 * q + g) ^ (2 * ((c16 & 7) >> 1)));
    // PERSISTENT: one workgroup per CU walks the tiles blockIdx.x, + gridDim.x, ... (a workgroup of this kernel owns its CU's LDS).  Consecutive
    // tiles of a workgroup mostly share a descriptor: it is copied to LDS when it changes.
    int di = 0, loaded = -1;
    for (int tile = (int)blockIdx.x; tile < total_tiles; tile += (int)gridDim.x) {
        while (di + 1 < n_desc && tile >= descs[di + 1].tile_start) ++di;
        if (di != loaded) {
            static_assert(sizeof(ComposeDesc) % 16 == 0, "descriptor copied in 16-byte pieces");
            __syncthreads();
            const u32x4* src = (const u32x4*)&descs[di];
            u32x4* dst = (u32x4*)&sdesc;
            for (int i = tid; i < (int)(sizeof(ComposeDesc) / 16); i += 512) dst[i] = src[i];
            __syncthreads();
            loaded = di;
        }
        // ---- the tile's scalars, once (locals: nothing below re-reads them from LDS)
        // (only the values of the stage loop are pinned in scalar registers; the epilogue's - base pointers, strides - are read from LDS
        // where they are used, as independent loads: pinning all of them overflowed the scalar file and the spills cost more than the reads)
#define w_ (sdesc.w)
#define cs_ (sdesc.col_scale)
#define ldw_ (sdesc.ldw)
#define ldo_ (sdesc.ldo)
#define nbs_ (sdesc.nb_stride)
#define nbo_ (sdesc.nb_offset)
        const int N_ = ct_u(sdesc.N), K_ = ct_u(sdesc.K), r_ = ct_u(sdesc.r), n_out_ = ct_u(sdesc.n_out);
        const int tiles_n_ = ct_u(sdesc.tiles_n), tiles_k_ = ct_u(sdesc.tiles_k), n_stages = ct_u(sdesc.n_stages);
        const int t_id = tile - ct_u(sdesc.tile_start);
        const int tn = t_id % tiles_n_, tk = t_id / tiles_n_;              // neighbours in the grid (and on an XCD: b, b + 8, ...) share the A^T tile
        const int n_tile = tn * CT_N, k_tile = tk * CT_K;
        const int n0 = n_tile + wn * 64, k0 = k_tile + wk * 32;
        const bool wave_on = k0 < K_;
        const int kblocks = K_ >> 5;
        const int np16 = (N_ + 15) & ~15;

        // ---- LDS-DMA sources of this wave: octets 2 wave, 2 wave + 1 of the A^T tile (rows k_tile + 8 o + row8) and of the B tile
        int at_off[2], b_off[2];                                          // elements (K r, N r < 2^31: host check)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            at_off[i] = min(k_tile + 8 * (wave * 2 + i) + row8, K_ - 1) * r_ + piece * 8;
            b_off[i] = min(n_tile + 8 * (wave * 2 + i) + row8, N_ - 1) * r_ + piece * 8;
        }
        auto issue = [&](int st) {                                         // stage st -> ring slot st % CT_RING: 4 DMA instructions per wave
            const bf16_t* at = ct_up(sdesc.st[st].at);
            const bf16_t* bm = ct_up(sdesc.st[st].bm);
            char* base = ct_smem + (st % CT_RING) * CT_HALF_BYTES;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_global_load_lds((ct_gbl_void*)(at + at_off[i]), (ct_lds_void*)(base + (wave * 2 + i) * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((ct_gbl_void*)(bm + b_off[i]), (ct_lds_void*)(base + (16 + wave * 2 + i) * 1024), 16, 0, 0);
            }
        };
#pragma unroll
        for (int st = 0; st < CT_RING - 1; ++st)
            if (st < n_stages) issue(st);

        // ---- W of the wave's 64 x 32 sub-tile, once, in the accumulator layout (see compose_multi_kernel::load_w)
        bf16x4 wv[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + i * 16 + c16;
            const int k = k0 + k8off;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (w_ && n < N_ && k < K_) v = *(const u32x4*)(w_ + (int64_t)n * ldw_ + k);
            auto r0 = __builtin_amdgcn_permlane16_swap(v[0], v[2], false, false);
            auto r1 = __builtin_amdgcn_permlane16_swap(v[1], v[3], false, false);
            const u32x2 lo2 = {r0[0], r1[0]}, hi2 = {r0[1], r1[1]};
            wv[i][0] = __builtin_bit_cast(bf16x4, lo2);
            wv[i][1] = __builtin_bit_cast(bf16x4, hi2);
        }
        float cs[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = k0 + t * 16 + g * 4;
            const f32x4 c4 = (cs_ && k < K_) ? *(const f32x4*)(cs_ + k) : (f32x4){1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[t][j] = c4[j];
        }

        f32x4 tot[4][2], acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t) { tot[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        float ret_num = 0.f, ret_den = 0.f;

        // the epilogue of row block i of output `o`: total (fp32) + W, column scale, ONE rounding, store (compose_multi_kernel::finish)
        auto finish = [&](const CtOut& o_, int i, const f32x4 (&tt)[2], const bf16x4 (&wvi)[2]) {
            const uint32_t seed = o_.seed;
            const bool want_ret = o_.ret != nullptr && w_ != nullptr;
            const int n = n0 + i * 16 + c16;
            const int nb = n >> 4;
            if (nb * 16 >= np16) return;
            const bool inb = n < N_;
            bf16x4 o[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int k = k0 + t * 16 + g * 4;
                float r4[4] = {tt[t][0], tt[t][1], tt[t][2], tt[t][3]};
                if (w_ && inb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) r4[j] += (float)wvi[t][j];
                }
                if (cs_) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) r4[j] *= cs[t][j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) o[t][j] = (inb && k + j < K_) ? compose_round(r4[j], seed, n, k + j) : (bf16_t)0.0f;
                if (want_ret && inb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < K_) {
                            const float d = tt[t][j] * cs[t][j];
                            const float moved = (float)o[t][j] - (float)(bf16_t)((float)wvi[t][j] * cs[t][j]);
                            ret_num = fmaf(moved, d, ret_num);
                            ret_den = fmaf(d, d, ret_den);
                        }
                }
            }
            const u32x2 pa = __builtin_bit_cast(u32x2, o[0]), pb = __builtin_bit_cast(u32x2, o[1]);
            auto r0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
            auto r1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
            const u32x4 ov = {r0[0], r1[0], r0[1], r1[1]};
            const int k8 = k0 + k8off;
            const int kb = k8 >> 5, q = (k8 & 31) >> 3;
            bf16_t* dst = o_.outp + ((int64_t)(nb * nbs_ + nbo_) * kblocks + kb) * 512 + ((q << 4) | (n & 15)) * 8;
            compose_store16(dst, ov, WT);
            if (o_.outr && inb && k8 < K_) *(u32x4*)(o_.outr + (int64_t)n * ldo_ + k8) = ov;
        };
        auto flush_ret = [&](int a, bool has) {                             // this output's retention sums: lanes, then one slot per wave
            if (has) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    ret_num += __shfl_xor(ret_num, o, 64);
                    ret_den += __shfl_xor(ret_den, o, 64);
                }
                if (lane == 0) { red[a][wave][0] = ret_num; red[a][wave][1] = ret_den; }
            }
            ret_num = 0.f; ret_den = 0.f;
        };

        // outputs without any term: W' = bf16(W c)
        for (int a = 0; a < n_out_; ++a) {
            if (ct_u((int)sdesc.out[a].mask) != 0) continue;
            const CtOut o_ = CtOut{ct_up(sdesc.out[a].outp), ct_up(sdesc.out[a].outr), ct_up(sdesc.out[a].ret), (uint32_t)ct_u((int)sdesc.out[a].seed), 0u};
            if (wave_on) {
                const f32x4 zero[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int i = 0; i < 4; ++i) finish(o_, i, zero, wv[i]);
            }
            flush_ret(a, o_.ret != nullptr);
        }

        for (int st = 0; st < n_stages; ++st) {
            // this wave's DMA of stage st has landed: at most the 4-DMA groups of the stages behind it may still be in flight (vmcnt counts in
            // issue order; the stores of an epilogue in between only make the wait a little stricter than needed)
            const int behind = min(CT_RING - 2, n_stages - 1 - st);
            if (behind >= 2) ct_wait_vm<8>();
            else if (behind == 1) ct_wait_vm<4>();
            else ct_wait_vm<0>();
            __syncthreads();                        // everyone's has, and everyone is done with stage st - 1: its ring slot is free
            if (st + CT_RING - 1 < n_stages) issue(st + CT_RING - 1);
            const float sc = __builtin_bit_cast(float, ct_u(__builtin_bit_cast(int, sdesc.st[st].scale)));
            const int flags = __builtin_amdgcn_readfirstlane(sdesc.st[st].flags);
            if (wave_on) {
                const bf16x8* hb = (const bf16x8*)(ct_smem + (st % CT_RING) * CT_HALF_BYTES);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    bf16x8 af[2], bf[4];
#pragma unroll
                    for (int t = 0; t < 2; ++t) af[t] = hb[(wk * 4 + t * 2) * 64 + fslot[q]];
#pragma unroll
                    for (int i = 0; i < 4; ++i) bf[i] = hb[(16 + wn * 8 + i * 2) * 64 + fslot[q]];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc[i][t] = mc_mfma_16x16x32(af[t], bf[i], acc[i][t]);
                }
            }
            // ranks beyond 128 are summed in chunks of 128 (one MFMA chain each), the chunk sums added in fp32 - as compose_multi_kernel
            if (flags & CT_CHUNK_END) {
                const bool first = (flags & CT_FIRST) != 0, last = (flags & CT_LAST) != 0;
                const int a = flags >> 8;
                CtOut o_{};
                if (last) o_ = CtOut{ct_up(sdesc.out[a].outp), ct_up(sdesc.out[a].outr), ct_up(sdesc.out[a].ret), (uint32_t)ct_u((int)sdesc.out[a].seed), 1u};
                if (wave_on) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const f32x4 base = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : tot[i][t];
                            tot[i][t] = base + acc[i][t] * sc;
                            acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                        if (last) finish(o_, i, tot[i], wv[i]);
                    }
                }
                if (last) flush_ret(a, o_.ret != nullptr);
            }
        }
        __syncthreads();
        // retention partials: one entry per (64-row half, 128-column tile): [2 tiles_n][tiles_k][2]
        if (tid < 2 * n_out_) {
            const int a = tid >> 1, half = tid & 1;
            float* ret = sdesc.out[a].ret;
            if (ret) {
                float* dst = ret + ((int64_t)(tn * 2 + half) * tiles_k_ + tk) * 2;
                const int w0 = half * 4;
                dst[0] = (red[a][w0][0] + red[a][w0 + 1][0]) + (red[a][w0 + 2][0] + red[a][w0 + 3][0]);
                dst[1] = (red[a][w0][1] + red[a][w0 + 1][1]) + (red[a][w0 + 2][1] + red[a][w0 + 3][1]);
            }
        }
        __syncthreads();                               // `red` and the ring are free for the next tile
#undef w_
#undef cs_
#undef ldw_
#undef ldo_
#undef nbs_
#undef nbo_
    }
}

#define MC_COMPOSE_NR 4

extern "C" int mc_compose_retention_floats(int N, int K, int64_t* floats) {
    MC_CHECK_ARG(floats && N > 0 && K > 0, "mc_compose_retention_floats: bad arguments");
    const int Kp = (K + 63) / 64 * 64;
    // the larger of the two kernels' layouts (general: [ceil(N / 64)][ceil(Kp / 256)], tile kernel: [2 ceil(N / 128)][ceil(Kp / 128)]); the caller
    // zero-fills the buffer and sums every entry
    const int64_t a = 2LL * ((Kp + 255) / 256) * ((N + 16 * MC_COMPOSE_NR - 1) / (16 * MC_COMPOSE_NR));
    const int64_t b = 2LL * ((Kp + CT_K - 1) / CT_K) * (2 * ((N + CT_N - 1) / CT_N));
    *floats = a > b ? a : b;
    return 0;
}

// argument block -> kernel parameters (validation shared by the single and the batched entry points)
static int compose_fill(const mc_compose_multi_args* a, ComposeMultiParams& p) {
    MC_CHECK_ARG(a && a->N > 0 && a->K > 0, "mc_compose_multi_bf16: bad arguments");
    MC_CHECK_ARG(a->n_terms >= 0 && a->n_terms <= MC_MAX_TERMS, "mc_compose_multi_bf16: at most %d terms (got %d)", MC_MAX_TERMS, a->n_terms);
    MC_CHECK_ARG(a->n_out >= 1 && a->n_out <= MC_MAX_OUTS, "mc_compose_multi_bf16: 1 .. %d outputs (got %d)", MC_MAX_OUTS, a->n_out);
    MC_CHECK_ARG(a->n_terms == 0 || (a->r > 0 && a->r % 32 == 0), "mc_compose_multi_bf16: rank %d must be a multiple of 32 (pad A^T / B)", a->r);
    MC_CHECK_ARG(!a->w || a->ldw % 4 == 0, "mc_compose_multi_bf16: ldw must be a multiple of 4");
    MC_CHECK_ARG(a->nb_stride >= 1 && a->nb_offset >= 0 && a->nb_offset < a->nb_stride, "mc_compose_multi_bf16: bad block interleave %d/%d", a->nb_offset, a->nb_stride);
    p.w = (const bf16_t*)a->w; p.ldw = a->ldw;
    for (int i = 0; i < MC_MAX_TERMS; ++i) { p.at[i] = nullptr; p.bm[i] = nullptr; p.scale[i] = 0.f; }
    for (int i = 0; i < a->n_terms; ++i) {
        MC_CHECK_ARG(a->at_list && a->b_list && a->scales && a->at_list[i] && a->b_list[i], "mc_compose_multi_bf16: null term %d", i);
        p.at[i] = (const bf16_t*)a->at_list[i]; p.bm[i] = (const bf16_t*)a->b_list[i]; p.scale[i] = a->scales[i];
    }
    p.n_terms = a->n_terms; p.r = a->r; p.n_out = a->n_out;
    for (int o = 0; o < MC_MAX_OUTS; ++o) {
        const bool on = o < a->n_out;
        MC_CHECK_ARG(!on || (a->out_packed && a->out_packed[o]), "mc_compose_multi_bf16: null output %d", o);
        MC_CHECK_ARG(!on || !a->term_mask || (a->term_mask[o] >> a->n_terms) == 0, "mc_compose_multi_bf16: output %d names a term beyond n_terms", o);
        p.out_packed[o] = on ? (bf16_t*)a->out_packed[o] : nullptr;
        p.out_rowmajor[o] = on && a->out_rowmajor ? (bf16_t*)a->out_rowmajor[o] : nullptr;
        p.term_mask[o] = on ? (a->term_mask ? a->term_mask[o] : ((1u << a->n_terms) - 1u)) : 0u;
        p.dither_seed[o] = on && a->dither_seeds ? a->dither_seeds[o] : 0u;
        p.retention[o] = on && a->retention_parts ? a->retention_parts[o] : nullptr;
    }
    p.ldo = a->ldo;
    p.N = a->N; p.K = a->K; p.Kp = (a->K + 63) / 64 * 64;
    p.col_scale = a->col_scale; p.nb_stride = a->nb_stride; p.nb_offset = a->nb_offset;
    return 0;
}

static bool compose_wt() {                              // MC_COMPOSE_WT=0: plain (write-back) output stores, for A/B
    static int wt = -1;
    if (wt < 0) { const char* e = getenv("MC_COMPOSE_WT"); wt = (e && e[0] == '0') ? 0 : 1; }
    return wt == 1;
}

// the 64 x 256 kernel (any shape; the tile kernel's fallback)
static void compose_launch_general(const ComposeMultiParams& p, hipStream_t s) {
    dim3 grid((p.Kp + 255) / 256, (p.N + 16 * MC_COMPOSE_NR - 1) / (16 * MC_COMPOSE_NR));
    bool fast = p.K % 8 == 0 && (!p.w || (p.ldw % 8 == 0 && ((uintptr_t)p.w & 15) == 0)) && (!p.col_scale || ((uintptr_t)p.col_scale & 15) == 0);
    for (int o = 0; o < p.n_out; ++o)
        if (p.out_rowmajor[o] && (p.ldo % 8 || ((uintptr_t)p.out_rowmajor[o] & 15))) fast = false;
    if (fast && compose_wt()) compose_multi_kernel<MC_COMPOSE_NR, true, true><<<grid, 256, 0, s>>>(p);
    else if (fast) compose_multi_kernel<MC_COMPOSE_NR, true, false><<<grid, 256, 0, s>>>(p);
    else compose_multi_kernel<MC_COMPOSE_NR, false, false><<<grid, 256, 0, s>>>(p);
}

// what compose_tile_kernel needs of a linear
static bool compose_tile_ok(const ComposeMultiParams& p) {
    if (p.K % 64 || (p.n_terms > 0 && p.r % 64)) return false;
    if ((int64_t)p.K * p.r >= (1LL << 31) || (int64_t)p.N * p.r >= (1LL << 31)) return false;
    if (p.w && (p.ldw % 8 || ((uintptr_t)p.w & 15))) return false;
    if (p.col_scale && ((uintptr_t)p.col_scale & 15)) return false;
    for (int m = 0; m < p.n_terms; ++m)
        if (((uintptr_t)p.at[m] & 15) || ((uintptr_t)p.bm[m] & 15)) return false;
    for (int o = 0; o < p.n_out; ++o)
        if (p.out_rowmajor[o] && (p.ldo % 8 || ((uintptr_t)p.out_rowmajor[o] & 15))) return false;
    static int off = -1;                                // MC_COMPOSE_TILE=0: the round-5 kernel for everything (A/B)
    if (off < 0) { const char* e = getenv("MC_COMPOSE_TILE"); off = (e && e[0] == '0') ? 1 : 0; }
    return off == 0;
}

#include <mutex>
#include <vector>
// Many linears, ONE launch of compose_tile_kernel (those that meet its conditions; the others take the general kernel, one launch each).
// The descriptor table lives in a library-owned device buffer that only grows (growing synchronises the device first).
extern "C" int mc_compose_batch_bf16(const mc_compose_multi_args* args, int n, void* stream) {
    MC_CHECK_ARG(args && n > 0, "mc_compose_batch_bf16: bad arguments");
    static std::mutex mu;
    static std::vector<ComposeDesc> host;
    static ComposeDesc* dev = nullptr;
    static size_t cap = 0;
    static hipEvent_t last_launch = nullptr;            // recorded behind the launch that reads the table: the next call waits for it
    std::lock_guard<std::mutex> lock(mu);
    hipStream_t s = (hipStream_t)stream;
    host.clear();
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        ComposeMultiParams q;
        const int rc = compose_fill(&args[i], q);
        if (rc) return rc;
        int n_pairs = 0;
        for (int o = 0; o < q.n_out; ++o) n_pairs += __builtin_popcount(q.term_mask[o]);
        const int hpt = q.n_terms > 0 ? q.r / 64 : 1;
        if (!compose_tile_ok(q) || n_pairs * hpt > CT_MAX_STAGES) { compose_launch_general(q, s); continue; }
        ComposeDesc d;
        memset(&d, 0, sizeof(d));
        d.w = q.w; d.col_scale = q.col_scale; d.ldw = q.ldw; d.ldo = q.ldo;
        d.N = q.N; d.K = q.K; d.r = q.r; d.n_out = q.n_out; d.nb_stride = q.nb_stride; d.nb_offset = q.nb_offset;
        d.tile_start = tiles;
        d.tiles_n = (q.N + CT_N - 1) / CT_N; d.tiles_k = (q.K + CT_K - 1) / CT_K;
        int ns = 0;
        for (int o = 0; o < q.n_out; ++o) {
            d.out[o] = CtOut{q.out_packed[o], q.out_rowmajor[o], q.retention[o], q.dither_seed[o], q.term_mask[o]};
            const uint32_t mask = q.term_mask[o];
            for (int m = 0; m < q.n_terms; ++m) {
                if (!((mask >> m) & 1u)) continue;
                for (int hh = 0; hh < hpt; ++hh) {
                    int fl = o << 8;
                    if ((hh & 1) == 1 || hh == hpt - 1) fl |= CT_CHUNK_END;
                    if (m == __builtin_ctz(mask) && hh <= 1) fl |= CT_FIRST;
                    if (m == 31 - __builtin_clz(mask) && hh == hpt - 1) fl |= CT_LAST;
                    d.st[ns++] = CtStage{q.at[m] + hh * 64, q.bm[m] + hh * 64, q.scale[m], fl, 0};
                }
            }
        }
        d.n_stages = ns;
        tiles += d.tiles_n * d.tiles_k;
        host.push_back(d);
    }
    if (!host.empty()) {
        // the previous batch's kernel (possibly on another stream) reads the table until it ends
        if (last_launch && hipEventSynchronize(last_launch) != hipSuccess) { mc_set_error("mc_compose_batch_bf16: %s", hipGetErrorString(hipGetLastError())); return 2; }
        if (host.size() > cap) {
            (void)hipDeviceSynchronize();
            if (dev) (void)hipFree(dev);
            dev = nullptr; cap = 0;
            const size_t want = host.size() < 256 ? 256 : host.size();
            if (hipMalloc((void**)&dev, want * sizeof(ComposeDesc)) != hipSuccess) { mc_set_error("mc_compose_batch_bf16: descriptor table allocation failed"); return 2; }
            cap = want;
        }
        hipError_t e = hipMemcpyAsync(dev, host.data(), host.size() * sizeof(ComposeDesc), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { mc_set_error("mc_compose_batch_bf16: descriptor copy: %s", hipGetErrorString(e)); return 2; }
        e = hipStreamSynchronize(s);                    // the pageable host table may be rewritten by the next call
        if (e != hipSuccess) { mc_set_error("mc_compose_batch_bf16: %s", hipGetErrorString(e)); return 2; }
        static bool attr = false;
        const int lds = CT_RING * CT_HALF_BYTES;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)compose_tile_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute((const void*)compose_tile_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr = true;
        }
        static int n_cu = 0;
        if (!n_cu) {
            int d = 0;
            hipDeviceProp_t pr;
            n_cu = (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
        }
        const int grid = tiles < n_cu ? tiles : n_cu;          // persistent: one workgroup per CU
        if (compose_wt()) compose_tile_kernel<true><<<grid, 512, lds, s>>>(dev, (int)host.size(), tiles);
        else compose_tile_kernel<false><<<grid, 512, lds, s>>>(dev, (int)host.size(), tiles);
        if (!last_launch && hipEventCreateWithFlags(&last_launch, hipEventDisableTiming) != hipSuccess) last_launch = nullptr;
        if (last_launch) (void)hipEventRecord(last_launch, s);
        else (void)hipStreamSynchronize(s);
    }
    MC_CHECK_LAUNCH();
    return 0;
}

extern "C" int mc_compose_multi_bf16(const mc_compose_multi_args* a, void* stream) {
    return mc_compose_batch_bf16(a, 1, stream);
}

extern "C" int mc_compose_weight_dither_bf16(const void* w, int64_t ldw, const void* const* at_list, const void* const* b_list,
                                             const float* scales, int n_terms, int r, void* out_packed, void* out_rowmajor,
                                             int64_t ldo, int N, int K, const float* col_scale, int nb_stride, int nb_offset,
                                             float* retention_parts, uint32_t dither_seed, void* stream) {
    MC_CHECK_ARG(out_packed && N > 0 && K > 0, "mc_compose_weight_bf16: bad arguments");
    mc_compose_multi_args a;
    void* outs[1] = {out_packed};
    void* rms[1] = {out_rowmajor};
    float* rets[1] = {retention_parts};
    uint32_t seeds[1] = {dither_seed};
    a.w = w; a.ldw = ldw; a.at_list = at_list; a.b_list = b_list; a.scales = scales; a.n_terms = n_terms; a.r = r;
    a.n_out = 1; a.out_packed = outs; a.out_rowmajor = rms; a.term_mask = nullptr; a.dither_seeds = seeds; a.retention_parts = rets;
    a.ldo = ldo; a.N = N; a.K = K; a.col_scale = col_scale; a.nb_stride = nb_stride; a.nb_offset = nb_offset;
    return mc_compose_multi_bf16(&a, stream);
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
