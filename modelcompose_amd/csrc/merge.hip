// TIES merging of task vectors on the GPU (HBM-bound integer / compare work over the flattened shared adapter tensors).
//
// Replaces scripts/model_composition/ties_merging.py:88-221 (reached through merge_unimodal_modelcompose.py:78-93,
// --strategy ties-{mean,sum,max}) for n checkpoints flattened to rows of x [n, d] (d ≈ 0.33 G for the rank-128 default adapters):
//   topk_values_mask (:88-109)  per row keep |x| >= the (d - int(d*K))-th smallest magnitude  ->  exact radix select on the float bits
//                               (3 histogram passes of 11 / 11 / 10 bits, LDS histograms merged with global atomics)
//   resolve_sign (:112-125)     sign of the column sum of the trimmed rows; zero sums take the majority sign
//   disjoint_merge (:128-157)   per column aggregate the trimmed entries whose sign agrees (mean / sum / max)
// Arithmetic mirrors torch's: fp32 accumulation in row order; for bf16 / fp16 checkpoints the column sum is rounded to the storage
// type once (torch.sum on a half tensor) and the mean divides that rounded sum by the fp32 count (bf16 / fp32 promotes to fp32).
#include "common.h"
#include <hip/hip_fp16.h>

template <typename T> __device__ __forceinline__ float ld_f(const T* p, int64_t i);
template <> __device__ __forceinline__ float ld_f<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_f<__bf16>(const __bf16* p, int64_t i) { return (float)p[i]; }
template <> __device__ __forceinline__ float ld_f<__half>(const __half* p, int64_t i) { return __half2float(p[i]); }
template <typename T> __device__ __forceinline__ float rnd_t(float v);
template <> __device__ __forceinline__ float rnd_t<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd_t<__bf16>(float v) { return (float)(__bf16)v; }
template <> __device__ __forceinline__ float rnd_t<__half>(float v) { return __half2float(__float2half(v)); }
template <typename T> __device__ __forceinline__ void st_t(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void st_t<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_t<__bf16>(__bf16* p, int64_t i, float v) { p[i] = (__bf16)v; }
template <> __device__ __forceinline__ void st_t<__half>(__half* p, int64_t i, float v) { p[i] = __float2half(v); }

#define TIES_BINS 2048

// hist[row][bin] += #{ j : (bits(|x[row][j]|) >> shift) & (nbins-1) == bin  and  bits >> pshift == prefix[row] (pshift < 32) }
template <typename T>
__global__ __launch_bounds__(256) void ties_hist_kernel(const T* __restrict__ x, int64_t ld, int64_t d, int shift, int nbins, const uint32_t* __restrict__ prefix,
                                                        int pshift, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[TIES_BINS];
    const int row = blockIdx.y;
    for (int i = threadIdx.x; i < nbins; i += 256) h[i] = 0;
    __syncthreads();
    const T* xr = x + (int64_t)row * ld;
    const uint32_t pre = pshift < 32 ? prefix[row] : 0u;
    for (int64_t j = blockIdx.x * 256LL + threadIdx.x; j < d; j += (int64_t)gridDim.x * 256) {
        const uint32_t b = __float_as_uint(fabsf(ld_f<T>(xr, j)));
        if (pshift >= 32 || (b >> pshift) == pre) atomicAdd(&h[(b >> shift) & (nbins - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += 256)
        if (h[i]) atomicAdd(&hist[(int64_t)row * TIES_BINS + i], h[i]);
}

// sign[j] = sgn(round_T(sum_i trimmed x[i][j]));  *sign_sum += sum_j sign[j]
template <typename T>
__global__ __launch_bounds__(256) void ties_sign_kernel(const T* __restrict__ x, int64_t ld, int64_t d, int n, const float* __restrict__ thr,
                                                        int8_t* __restrict__ sign, long long* __restrict__ sign_sum) {
    __shared__ float red[16];
    long long local = 0;
    for (int64_t j = blockIdx.x * 256LL + threadIdx.x; j < d; j += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) {
            const float v = ld_f<T>(x + (int64_t)i * ld, j);
            s += fabsf(v) >= thr[i] ? v : 0.f;
        }
        s = rnd_t<T>(s);
        const int sg = (s > 0.f) - (s < 0.f);
        sign[j] = (int8_t)sg;
        local += sg;
    }
    // block reduction (|local| per thread is small enough for an exact float sum)
    const float tot = block_sum((float)local, red);
    if (threadIdx.x == 0 && tot != 0.f) atomicAdd((unsigned long long*)sign_sum, (unsigned long long)(long long)tot);
}

// func: 0 mean, 1 sum, 2 max
template <typename T>
__global__ __launch_bounds__(256) void ties_merge_kernel(const T* __restrict__ x, int64_t ld, int64_t d, int n, const float* __restrict__ thr,
                                                         const int8_t* __restrict__ sign, const long long* __restrict__ sign_sum, int func,
                                                         T* __restrict__ out) {
    const long long ss = *sign_sum;
    const int majority = (ss > 0) - (ss < 0);
    for (int64_t j = blockIdx.x * 256LL + threadIdx.x; j < d; j += (int64_t)gridDim.x * 256) {
        int sg = sign[j];
        if (sg == 0) sg = majority;
        float sum = 0.f, mx = 0.f;
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            float v = ld_f<T>(x + (int64_t)i * ld, j);
            v = fabsf(v) >= thr[i] ? v : 0.f;
            const bool keep = sg > 0 ? v > 0.f : v < 0.f;       // ties_merging.py:135-137 (an all-zero election keeps the negative entries)
            const float sel = keep ? v : 0.f;
            sum += sel;
            cnt += sel != 0.f;
            mx = fmaxf(mx, fabsf(sel));
        }
        float r;
        if (func == 0) r = rnd_t<T>(sum) / (float)max(cnt, 1);
        else if (func == 1) r = rnd_t<T>(sum);
        else r = mx * (float)sg;
        st_t<T>(out, j, r);
    }
}

template <typename T>
static int ties_launch(int which, const void* x, int64_t ld, int64_t d, int n, int a0, int a1, const uint32_t* prefix, int pshift, uint32_t* hist,
                       const float* thr, int8_t* sign, long long* sign_sum, int func, void* out, hipStream_t s) {
    const int blocks = (int)min((int64_t)4096, (d + 255) / 256);
    if (which == 0) ties_hist_kernel<T><<<dim3(blocks, n), 256, 0, s>>>((const T*)x, ld, d, a0, a1, prefix, pshift, hist);
    else if (which == 1) ties_sign_kernel<T><<<blocks, 256, 0, s>>>((const T*)x, ld, d, n, thr, sign, sign_sum);
    else ties_merge_kernel<T><<<blocks, 256, 0, s>>>((const T*)x, ld, d, n, thr, sign, sign_sum, func, (T*)out);
    return 0;
}

#define TIES_DISPATCH(...)                                                                                  \
    do {                                                                                                    \
        if (dtype == MC_DTYPE_F32) ties_launch<float>(__VA_ARGS__);                                         \
        else if (dtype == MC_DTYPE_BF16) ties_launch<__bf16>(__VA_ARGS__);                                    \
        else if (dtype == MC_DTYPE_F16) ties_launch<__half>(__VA_ARGS__);                                   \
        else { mc_set_error("ties: unsupported dtype code %d", dtype); return 1; }                          \
    } while (0)

// one histogram pass of the radix select; hist [n][2048] must be zeroed by the caller
extern "C" int mc_ties_hist(const void* x, int dtype, int64_t ld, int64_t d, int n, int shift, int nbins, const uint32_t* prefix, int prefix_shift,
                            uint32_t* hist, void* stream) {
    MC_CHECK_ARG(x && hist && d > 0 && n > 0 && nbins > 0 && nbins <= TIES_BINS && (nbins & (nbins - 1)) == 0, "mc_ties_hist: bad arguments");
    MC_CHECK_ARG(prefix_shift >= 32 || prefix, "mc_ties_hist: prefix missing");
    TIES_DISPATCH(0, x, ld, d, n, shift, nbins, prefix, prefix_shift, hist, nullptr, nullptr, nullptr, 0, nullptr, (hipStream_t)stream);
    MC_CHECK_LAUNCH();
    return 0;
}

// thr [n] fp32 = per-row trim threshold (the k-th smallest magnitude); sign [d] int8 and sign_sum (int64, zeroed by the caller) are
// scratch / outputs of the sign election; out [d] in the input dtype.  func: 0 mean, 1 sum, 2 max.
extern "C" int mc_ties_merge(const void* x, int dtype, int64_t ld, int64_t d, int n, const float* thr, int8_t* sign, long long* sign_sum, int func,
                             void* out, void* stream) {
    MC_CHECK_ARG(x && thr && sign && sign_sum && out && d > 0 && n > 0 && func >= 0 && func <= 2, "mc_ties_merge: bad arguments");
    TIES_DISPATCH(1, x, ld, d, n, 0, 0, nullptr, 32, nullptr, thr, sign, sign_sum, func, out, (hipStream_t)stream);
    TIES_DISPATCH(2, x, ld, d, n, 0, 0, nullptr, 32, nullptr, thr, sign, sign_sum, func, out, (hipStream_t)stream);
    MC_CHECK_LAUNCH();
    return 0;
}

// ---- parameter-interference metrics of a merged checkpoint (scripts/model_composition/calculate_metrics.py:26-37, :62-67) ----
// One pass over the n task vectors accumulates, per workgroup and in double precision,
//   [0] sum (x0-x1)^2   [1] sum x0*x1   [2] sum x0^2   [3] sum x1^2          (L2 :26-27 and cosine :29-30 use rows 0 and 1 only)
//   [4] sum_j |sum_i x_ij| / sum_i |x_ij|  over columns with sum_i |x_ij| != 0,  [5] the number of such columns   (SSD :32-37)
//   [6], [7] the same two with every row trimmed to |x| >= thr[i] (TSSD = SSD of topk_values_mask(K=50), :61-64)
// partial [gridDim.x][8] is summed on the host in block order, so the result does not depend on scheduling.
#define METRIC_BLOCKS 1024
template <typename T>
__global__ __launch_bounds__(256) void merge_metrics_kernel(const T* __restrict__ x, int64_t ld, int64_t d, int n, const float* __restrict__ thr,
                                                            double* __restrict__ partial) {
    __shared__ double red[8][4];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t j = blockIdx.x * 256LL + threadIdx.x; j < d; j += (int64_t)gridDim.x * 256) {
        float s = 0.f, a = 0.f, st = 0.f, at = 0.f, x0 = 0.f, x1 = 0.f;
        for (int i = 0; i < n; ++i) {
            const float v = ld_f<T>(x + (int64_t)i * ld, j);
            if (i == 0) x0 = v;
            if (i == 1) x1 = v;
            s += v;
            a += fabsf(v);
            const float t = (thr && fabsf(v) >= thr[i]) ? v : 0.f;
            st += t;
            at += fabsf(t);
        }
        const float df = x0 - x1;
        acc[0] += (double)(df * df);
        acc[1] += (double)(x0 * x1);
        acc[2] += (double)(x0 * x0);
        acc[3] += (double)(x1 * x1);
        if (a != 0.f) { acc[4] += (double)fabsf(s / a); acc[5] += 1.0; }
        if (at != 0.f) { acc[6] += (double)fabsf(st / at); acc[7] += 1.0; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) partial[(int64_t)blockIdx.x * 8 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// x [n][d] (n >= 2) task vectors; thr [n] fp32 trim thresholds or NULL (then [6],[7] are zero); partial: METRIC_BLOCKS*8 doubles.
extern "C" int mc_merge_metrics(const void* x, int dtype, int64_t ld, int64_t d, int n, const float* thr, double* partial, void* stream) {
    MC_CHECK_ARG(x && partial && d > 0 && n >= 2, "mc_merge_metrics: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MC_DTYPE_F32) merge_metrics_kernel<float><<<METRIC_BLOCKS, 256, 0, s>>>((const float*)x, ld, d, n, thr, partial);
    else if (dtype == MC_DTYPE_BF16) merge_metrics_kernel<__bf16><<<METRIC_BLOCKS, 256, 0, s>>>((const __bf16*)x, ld, d, n, thr, partial);
    else if (dtype == MC_DTYPE_F16) merge_metrics_kernel<__half><<<METRIC_BLOCKS, 256, 0, s>>>((const __half*)x, ld, d, n, thr, partial);
    else { mc_set_error("mc_merge_metrics: unsupported dtype code %d", dtype); return 1; }
    MC_CHECK_LAUNCH();
    return 0;
}
extern "C" int mc_merge_metrics_blocks(void) { return METRIC_BLOCKS; }
