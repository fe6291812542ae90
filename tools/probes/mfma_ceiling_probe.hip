// What bounds an LDS-fed MFMA loop on this part: the matrix pipe, the LDS fragment reads, or the power the two draw together?
// Variants of the 256 x 256 x 64 workgroup K-step of gemm_tile256_kernel (no global traffic, random bf16 operands resident in LDS):
//   w8r   8 waves x (128 x 64) per wave, fragments re-read from LDS every K-step (24 KiB per wave-step: the production kernel's MFMA halves)
//   w4r   4 waves x (128 x 128) per wave (256 accumulator registers, one wave per SIMD), fragments register double-buffered
//         (32 KiB per wave-step: 2/3 of w8r's LDS bytes per FLOP)
//   w8n   8 waves, fragments read ONCE before the loop: the matrix pipe alone
// each for v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16.  Reports wall TFLOP/s and the in-kernel clock.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/mfma_ceiling_probe tools/probes/mfma_ceiling_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KSTEPS 4096

enum { W8R = 0, W4R = 1, W8N = 2 };

// LDS image: 2 K-steps x (W 32 KiB | X 32 KiB), 1-KiB lane-linear fragment blocks.
//   16x16x32: block (rb16, kk) at ((rb16 * 2 + kk) * 1024), rb16 < 16, kk < 2
//   32x32x16: block (rb32, k4) at ((rb32 * 4 + k4) * 1024), rb32 < 8,  k4 < 4
template <int MODE, int SHAPE>
__global__ __launch_bounds__(MODE == W4R ? 256 : 512, MODE == W4R ? 1 : 2) void probe(const bf16x8* __restrict__ init, float* __restrict__ out,
                                                                                       unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = MODE == W4R ? 256 : 512;
    for (int i = tid; i < 8192; i += NT) ((bf16x8*)smem)[i] = init[(blockIdx.x * 8192 + i) & 65535];
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if constexpr (MODE == W4R) {
        const int wave_n = wave >> 1, wave_m = wave & 1;
        if constexpr (SHAPE == 16) {
            f32x4 acc[8][8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16x8 wf[2][8], xf[2][8];
            auto load = [&](int buf, int ks, int kk) {
                const char* base = smem + (ks & 1) * 65536;
#pragma unroll
                for (int i = 0; i < 8; ++i) wf[buf][i] = *(const bf16x8*)(base + ((wave_n * 8 + i) * 2 + kk) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) xf[buf][j] = *(const bf16x8*)(base + 32768 + ((wave_m * 8 + j) * 2 + kk) * 1024 + lane * 16);
            };
            load(0, 0, 0);
            for (int ks = 0; ks < KSTEPS; ++ks) {
                asm volatile("" ::: "memory");
                load(1, ks, 1);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], xf[0][j], acc[i][j], 0, 0, 0);
                asm volatile("" ::: "memory");
                load(0, ks + 1, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][i], xf[1][j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        } else {
            f32x16 acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            bf16x8 wf[2][4], xf[2][4];
            auto load = [&](int buf, int ks, int k4) {
                const char* base = smem + (ks & 1) * 65536;
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[buf][i] = *(const bf16x8*)(base + ((wave_n * 4 + i) * 4 + k4) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[buf][j] = *(const bf16x8*)(base + 32768 + ((wave_m * 4 + j) * 4 + k4) * 1024 + lane * 16);
            };
            load(0, 0, 0);
            for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    asm volatile("" ::: "memory");
                    if (k4 < 3) load((k4 + 1) & 1, ks, k4 + 1); else load(0, ks + 1, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k4 & 1][i], xf[k4 & 1][j], acc[i][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        }
    } else {
        const int wave_n = wave >> 2, wave_m = wave & 3;
        if constexpr (SHAPE == 16) {
            f32x4 acc[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            bf16x8 wf[2][8], xf[2][4];
            auto load = [&](int kk, const char* base) {
#pragma unroll
                for (int i = 0; i < 8; ++i) wf[kk][i] = *(const bf16x8*)(base + ((wave_n * 8 + i) * 2 + kk) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[kk][j] = *(const bf16x8*)(base + 32768 + ((wave_m * 4 + j) * 2 + kk) * 1024 + lane * 16);
            };
            if constexpr (MODE == W8N) { load(0, smem); load(1, smem); }
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const char* base = smem + (ks & 1) * 65536;
                if constexpr (MODE == W8R) asm volatile("" ::: "memory");
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    if constexpr (MODE == W8R) load(kk, base);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][i], xf[kk][j], acc[i][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        } else {
            f32x16 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            bf16x8 wf[4][4], xf[4][2];
            auto load = [&](int k4, const char* base) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[k4][i] = *(const bf16x8*)(base + ((wave_n * 4 + i) * 4 + k4) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[k4][j] = *(const bf16x8*)(base + 32768 + ((wave_m * 2 + j) * 4 + k4) * 1024 + lane * 16);
            };
            if constexpr (MODE == W8N) { load(0, smem); load(1, smem); load(2, smem); load(3, smem); }
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const char* base = smem + (ks & 1) * 65536;
                if constexpr (MODE == W8R) asm volatile("" ::: "memory");
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    if constexpr (MODE == W8R) load(k4, base);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[k4][i], xf[k4][j], acc[i][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        }
    }
    out[blockIdx.x * 512 + tid] = s;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef void (*kern_t)(const bf16x8*, float*, unsigned long long*);

int main(int argc, char** argv) {
    const double warm_s = argc > 1 ? atof(argv[1]) : 2.0;
    const int blocks = 256;
    std::vector<uint16_t> h(65536 * 8);
    uint32_t st = 12345u;
    for (auto& v : h) {
        st = st * 1664525u + 1013904223u;
        const uint32_t r = st >> 8;
        v = (uint16_t)(((r & 1) << 15) | ((120 + (r >> 1) % 7) << 7) | ((r >> 8) & 127));
    }
    bf16x8* init; float* out; unsigned long long* stamps;
    CK(hipMalloc(&init, h.size() * 2)); CK(hipMalloc(&out, blocks * 512 * 4)); CK(hipMalloc(&stamps, blocks * 16));
    CK(hipMemcpy(init, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    struct V { const char* name; kern_t k; int threads; } vs[] = {
        {"w8r_16x16x32", probe<W8R, 16>, 512}, {"w8r_32x32x16", probe<W8R, 32>, 512},
        {"w4r_16x16x32", probe<W4R, 16>, 256}, {"w4r_32x32x16", probe<W4R, 32>, 256},
        {"w8n_16x16x32", probe<W8N, 16>, 512}, {"w8n_32x32x16", probe<W8N, 32>, 512},
    };
    for (auto& v : vs) CK(hipFuncSetAttribute((const void*)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const V& v, int reps, double* ms, double* clk_ghz) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) v.k<<<blocks, v.threads, 131072>>>(init, out, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        *ms = t / reps;
        std::vector<unsigned long long> hs(blocks * 2);
        CK(hipMemcpy(hs.data(), stamps, blocks * 16, hipMemcpyDeviceToHost));
        std::vector<double> clk;
        for (int b = 0; b < blocks; ++b) clk.push_back((double)hs[2 * b] / (double)hs[2 * b + 1] * 0.1);
        std::sort(clk.begin(), clk.end());
        *clk_ghz = clk[blocks / 2];
    };
    const double flop = 2.0 * 256 * 256 * 64 * (double)KSTEPS * blocks;
    double ms, clk;
    for (double spent = 0; spent < warm_s * 1e3;) for (auto& v : vs) { run(v, 4, &ms, &clk); spent += 4 * ms; }
    printf("{\"probe\": \"mfma_ceiling\", \"ksteps\": %d, \"rounds\": [", KSTEPS);
    bool first = true;
    for (int round = 0; round < 4; ++round)
        for (auto& v : vs) {
            run(v, 8, &ms, &clk);
            printf("%s{\"variant\": \"%s\", \"ms\": %.4f, \"tflops\": %.1f, \"clock_ghz\": %.3f}", first ? "" : ", ", v.name, ms, flop / (ms * 1e-3) / 1e12, clk);
            first = false;
        }
    printf("]}\n");
    return 0;
}
