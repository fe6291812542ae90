// Sampled decoding step: temperature -> top-k -> top-p -> multinomial, one 1024-thread workgroup per batch row.
//
// The reference reaches this through model.generate(do_sample=True, temperature, top_p, ...) (eval/model_multimodal_qa_loader.py:94-102,
// default --temperature 0.2 at :135; serve/model_worker.py:160-185), i.e. the logits warpers of transformers==4.31.0
// generation/logits_process.py (third-party, absent from /root/reference; restated in oracle/sampling.py and pinned there against the
// installed transformers' own warper classes):
//   TemperatureLogitsWarper  s = logits / T
//   TopKLogitsWarper         keep s >= (k-th largest s)            (ties at the k-th value are kept; GenerationConfig default top_k = 50)
//   TopPLogitsWarper         ascending order, drop while the inclusive cumulative softmax mass <= 1 - top_p; the largest is always kept
//   sample                   multinomial(softmax(kept))
// Everything after the softmax runs on integer masses q_i = floor(p_i * 2^40) so that the cumulative sums are exact and independent of
// the order of accumulation: the kept set and the drawn token are bitwise reproducible for a given (seed, row, step).
// The k-th largest score and the top-p cut are both found by 3-pass radix selects over the float bits (11 + 11 + 10), counts for top-k
// and masses for top-p; the logits row (128 KB at vocab 32000) stays in L2 between passes.
// Deviations from torch: probabilities equal to the top-p cut value are all kept (torch removes an arbitrary subset of such a tie group),
// and kept tokens whose probability is below 2^-40 carry no mass (never drawn).
#include "common.h"

#define SMP_THREADS 1024
#define SMP_BINS 2048
#define SMP_FIX 1099511627776.0f /* 2^40 */

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t ord_key(float v) {       // monotone float -> uint (ascending)
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_val(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Philox4x32-10 (Salmon et al. 2011); counter = (row, step, 0, 0), key = seed
__device__ __forceinline__ uint32_t philox_u32(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1) {
    uint32_t c2 = 0, c3 = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

// exclusive prefix sum over the workgroup's threads; returns the exclusive value, *total = sum over all threads.  scratch: 17 u64 of LDS
__device__ __forceinline__ u64 block_exscan(u64 v, u64* scratch, u64* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u64 t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (int w = 0; w < SMP_THREADS / 64; ++w) { const u64 t = scratch[w]; scratch[w] = run; run += t; }
        scratch[16] = run;
    }
    __syncthreads();
    *total = scratch[16];
    return scratch[wave] + inc - v;
}

struct SampleParams {
    const float* logits; int64_t ld;
    int64_t* next_ids; int64_t* out_ids; int64_t ld_out;
    const int32_t* step_ptr;       // device step counter (column of out_ids, RNG counter); NULL = step_const
    const uint32_t* seed_ptr;      // device seed (2 x u32); NULL = seed_const
    int N; float temp; int top_k; float top_p;
    u64 seed_const; int step_const;
    const float* uniform_in;       // optional [M] uniforms in [0,1) replacing the RNG (tests)
    float* probs_out; int64_t ldp; // optional [M][N] final probabilities (0 for removed tokens)
};

// One radix-select pass: histogram (counts or masses) of digit `shift/bits` over the elements whose higher key bits equal `prefix`, then
// walk the bins from the top (descending = 1) or the bottom to the bin in which the running total first exceeds `budget`.
// Returns that bin; *before = total of the bins walked past.  All threads get the same results.
template <bool MASS, bool DESC, typename F>
__device__ __forceinline__ int radix_pass(F&& key_and_weight, int N, int shift, int nbins, uint32_t prefix, int pshift, u64 budget, u64* hist,
                                          u64* scratch, u64* before) {
    for (int i = threadIdx.x; i < nbins; i += SMP_THREADS) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += SMP_THREADS) {
        uint32_t k; u64 w;
        key_and_weight(i, k, w);
        if (w && (pshift >= 32 || (k >> pshift) == prefix)) atomicAdd(&hist[(k >> shift) & (nbins - 1)], MASS ? w : (u64)1);
    }
    __syncthreads();
    // two bins per thread, in walk order
    const int b0 = 2 * threadIdx.x, b1 = b0 + 1;
    const int i0 = DESC ? nbins - 1 - b0 : b0, i1 = DESC ? nbins - 1 - b1 : b1;
    const u64 h0 = b0 < nbins ? hist[i0] : 0, h1 = b1 < nbins ? hist[i1] : 0;
    u64 total;
    const u64 ex = block_exscan(h0 + h1, scratch, &total);
    __shared__ int s_bin;
    __shared__ u64 s_before;
    if (threadIdx.x == 0) { s_bin = -1; s_before = total; }
    __syncthreads();
    if (h0 && ex <= budget && ex + h0 > budget) { s_bin = i0; s_before = ex; }
    else if (h1 && ex + h0 <= budget && ex + h0 + h1 > budget) { s_bin = i1; s_before = ex + h0; }
    __syncthreads();
    *before = s_before;
    return s_bin;
}

__global__ __launch_bounds__(SMP_THREADS) void sample_step_kernel(SampleParams p) {
    __shared__ u64 hist[SMP_BINS];
    __shared__ u64 scratch[17];
    __shared__ float redf[16];
    const int row = blockIdx.x, N = p.N;
    const float* lr = p.logits + (int64_t)row * p.ld;
    auto score = [&](int i) { return lr[i] / p.temp; };       // IEEE fp32 division, as torch's `scores / temperature`

    // ---- top-k threshold (key of the k-th largest score); keep key >= kth
    uint32_t kth = 0;
    if (p.top_k > 0 && p.top_k < N) {
        u64 budget = (u64)(p.top_k - 1);          // walk descending until the running count exceeds k-1
        uint32_t prefix = 0;
        const int shifts[3] = {21, 10, 0}, nb[3] = {2048, 2048, 1024}, ps[3] = {32, 21, 10};
        for (int pass = 0; pass < 3; ++pass) {
            u64 before;
            const int bin = radix_pass<false, true>([&](int i, uint32_t& k, u64& w) { k = ord_key(score(i)); w = 1; }, N, shifts[pass], nb[pass],
                                                    prefix, ps[pass], budget, hist, scratch, &before);
            budget -= before;
            prefix = (prefix << (pass == 2 ? 10 : 11)) | (uint32_t)bin;
        }
        kth = prefix;
    }
    // ---- softmax over the kept scores
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < N; i += SMP_THREADS) mx = fmaxf(mx, score(i));
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = mx;
    __syncthreads();
    for (int w = 0; w < SMP_THREADS / 64; ++w) mx = fmaxf(mx, redf[w]);
    float se = 0.f;
    for (int i = threadIdx.x; i < N; i += SMP_THREADS) {
        const float s = score(i);
        if (ord_key(s) >= kth) se += __expf(s - mx);
    }
    se = block_sum(se, redf);
    const float inv = 1.0f / se;
    auto prob = [&](int i) -> float {                          // softmax probability; 0 for tokens cut by top-k
        const float s = score(i);
        return ord_key(s) >= kth ? __expf(s - mx) * inv : 0.f;
    };
    auto mass = [&](int i) -> u64 { return (u64)(prob(i) * SMP_FIX); };     // q_i = floor(p_i * 2^40)
    // ---- top-p cut: ascending by probability, drop while inclusive cumulative mass <= (1 - top_p)
    uint32_t cut = 0;                                           // keep float bits of p >= cut
    u64 total_q = 0;
    {
        u64 part = 0;
        for (int i = threadIdx.x; i < N; i += SMP_THREADS) part += mass(i);
        (void)block_exscan(part, scratch, &total_q);
    }
    if (p.top_p < 1.0f) {
        const float drop = 1.0f - p.top_p;
        u64 budget = (u64)((double)drop * (double)total_q);     // masses sum to total_q (slightly under 2^40 after flooring)
        uint32_t prefix = 0;
        bool found = true;
        const int shifts[3] = {21, 10, 0}, nb[3] = {2048, 2048, 1024}, ps[3] = {32, 21, 10};
        for (int pass = 0; pass < 3 && found; ++pass) {
            u64 before;
            const int bin = radix_pass<true, false>(
                [&](int i, uint32_t& k, u64& w) { const float pr = prob(i); k = __float_as_uint(pr); w = (u64)(pr * SMP_FIX); },
                N, shifts[pass], nb[pass], prefix, ps[pass], budget, hist, scratch, &before);
            if (bin < 0) { found = false; break; }
            budget -= before;
            prefix = (prefix << (pass == 2 ? 10 : 11)) | (uint32_t)bin;
        }
        cut = found ? prefix : 0xffffffffu;
    }
    auto kept_mass = [&](int i) -> u64 {
        const float pr = prob(i);
        return __float_as_uint(pr) >= cut ? (u64)(pr * SMP_FIX) : 0;
    };
    // the largest probability always survives (min_tokens_to_keep = 1): if the cut removed everything, keep the arg-max tokens
    u64 zq = 0;
    {
        u64 part = 0;
        for (int i = threadIdx.x; i < N; i += SMP_THREADS) part += kept_mass(i);
        (void)block_exscan(part, scratch, &zq);
    }
    const bool only_max = zq == 0;
    auto final_mass = [&](int i) -> u64 {
        if (!only_max) return kept_mass(i);
        return score(i) == mx ? (u64)1 : (u64)0;
    };
    if (only_max) {
        u64 part = 0;
        for (int i = threadIdx.x; i < N; i += SMP_THREADS) part += final_mass(i);
        (void)block_exscan(part, scratch, &zq);
    }
    // ---- multinomial by inverse CDF in index order (contiguous chunk per thread)
    float u;
    if (p.uniform_in) u = p.uniform_in[row];
    else {
        const int step = p.step_ptr ? *p.step_ptr : p.step_const;
        const uint32_t k0 = p.seed_ptr ? p.seed_ptr[0] : (uint32_t)p.seed_const, k1 = p.seed_ptr ? p.seed_ptr[1] : (uint32_t)(p.seed_const >> 32);
        u = (float)(philox_u32((uint32_t)row, (uint32_t)step, k0, k1) >> 8) * (1.0f / 16777216.0f);
    }
    u64 target = (u64)((double)u * (double)zq);
    if (target >= zq) target = zq - 1;
    const int chunk = (N + SMP_THREADS - 1) / SMP_THREADS;
    const int lo = threadIdx.x * chunk, hi = min(lo + chunk, N);
    u64 part = 0;
    for (int i = lo; i < hi; ++i) part += final_mass(i);
    u64 tot;
    const u64 ex = block_exscan(part, scratch, &tot);
    if (part && ex <= target && ex + part > target) {
        u64 run = ex;
        int pick = lo;
        for (int i = lo; i < hi; ++i) {
            const u64 q = final_mass(i);
            if (q && run + q > target) { pick = i; break; }
            run += q;
        }
        p.next_ids[row] = pick;
        if (p.out_ids) p.out_ids[row * p.ld_out + (p.step_ptr ? *p.step_ptr : p.step_const)] = pick;
    }
    if (p.probs_out) {
        const double z = (double)zq;
        for (int i = threadIdx.x; i < N; i += SMP_THREADS) p.probs_out[(int64_t)row * p.ldp + i] = (float)((double)final_mass(i) / z);
    }
}

extern "C" int mc_sample_step_f32(const float* logits, int64_t ld, int64_t* next_ids, int64_t* out_ids, int64_t ld_out, const int32_t* step_ptr,
                                  int step_const, const uint32_t* seed_ptr, unsigned long long seed_const, int M, int N, float temperature,
                                  int top_k, float top_p, const float* uniform_in, float* probs_out, int64_t ldp, void* stream) {
    MC_CHECK_ARG(logits && next_ids && M > 0 && N > 0, "mc_sample_step_f32: bad arguments");
    MC_CHECK_ARG(temperature > 0.f, "mc_sample_step_f32: `temperature` (=%g) has to be a strictly positive float", (double)temperature);
    MC_CHECK_ARG(top_p >= 0.f && top_p <= 1.f, "mc_sample_step_f32: `top_p` has to be a float > 0 and < 1, but is %g", (double)top_p);
    MC_CHECK_ARG(top_k >= 0, "mc_sample_step_f32: `top_k` has to be a positive integer, but is %d", top_k);
    SampleParams p;
    p.logits = logits; p.ld = ld; p.next_ids = next_ids; p.out_ids = out_ids; p.ld_out = ld_out; p.step_ptr = step_ptr; p.seed_ptr = seed_ptr;
    p.N = N; p.temp = temperature; p.top_k = top_k; p.top_p = top_p; p.seed_const = seed_const; p.step_const = step_const;
    p.uniform_in = uniform_in; p.probs_out = probs_out; p.ldp = ldp;
    sample_step_kernel<<<M, SMP_THREADS, 0, (hipStream_t)stream>>>(p);
    MC_CHECK_LAUNCH();
    return 0;
}


// ------------------------------------------------------------------------------------------
// log_softmax over fp32 rows: the scoring step of beam search (transformers 4.31 generation/utils.py beam_search:
// next_token_scores = log_softmax(next_token_logits) - eval/model_multimodal_qa_loader.py:94-102 forwards --num_beams).  One workgroup per
// row, two sweeps (max, sum of exp) then the write; fp32 throughout, fixed reduction order.
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ x, int64_t ld, float* __restrict__ out, int64_t ldo, int N) {
    __shared__ float red[16];
    const float* r = x + (int64_t)blockIdx.x * ld;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < N; i += 256) m = fmaxf(m, r[i]);
    m = wave_max(m);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) s += __expf(r[i] - m);
    const float tot = block_sum(s, red + 4);
    const float lse = m + __logf(tot);
    float* o = out + (int64_t)blockIdx.x * ldo;
    for (int i = threadIdx.x; i < N; i += 256) o[i] = r[i] - lse;
}

extern "C" int mc_log_softmax_f32(const float* logits, int64_t ld, float* out, int64_t ldo, int M, int N, void* stream) {
    MC_CHECK_ARG(logits && out && M > 0 && N > 0, "mc_log_softmax_f32: bad arguments");
    log_softmax_kernel<<<M, 256, 0, (hipStream_t)stream>>>(logits, ld, out, ldo, N);
    MC_CHECK_LAUNCH();
    return 0;
}
