// A/B of the two bf16 MFMA shapes at the SAME per-wave output tile (128 x 64, the tile of gemm_tile256_kernel's waves),
// operands re-read from LDS by ds_read_b128 every K-step, random data, two waves per SIMD (512-thread workgroups, one per CU) - the
// regime of the production GEMM's MFMA halves.  Reports wall TFLOP/s, the in-kernel clock (s_memtime / s_memrealtime) and cycles per
// K-step for v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16.  MI355X_MICROARCH.md "DVFS give-back" item 7 / guide rule 28: the
// shape the chip clocks higher wins by wall, cycles per FLOP do not decide.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/mfma_shape_probe tools/probes/mfma_shape_probe.hip
//   ./tools/probes/mfma_shape_probe [seconds_of_warmup]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KSTEPS 4096          // K-steps of 64 per workgroup (one "tile" = 256 x 256 x 64: 2048 MFMA cycles per SIMD)

// LDS image per K-step: W 256 rows x 64 k and X 256 rows x 64 k in 1-KiB fragment blocks, lane-linear (conflict-free for both shapes'
// reads as laid out below); the same two K-step images are re-read (no global traffic: this isolates LDS-read + MFMA)
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void probe(const bf16x8* __restrict__ init, float* __restrict__ out, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = wave >> 2, wave_m = wave & 3;
    // fill 128 KiB of LDS with random bf16 (2 K-steps x (W 32 KiB | X 32 KiB))
    for (int i = tid; i < 8192; i += 512) ((bf16x8*)smem)[i] = init[(blockIdx.x * 8192 + i) & 65535];
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 16) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const char* base = smem + (ks & 1) * 65536;
            asm volatile("" ::: "memory");               // the fragments are re-read every K-step, as a GEMM's are
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 wf[8], xf[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) wf[i] = *(const bf16x8*)(base + ((wave_n * 8 + i) * 2 + kk) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[j] = *(const bf16x8*)(base + 32768 + ((wave_m * 4 + j) * 2 + kk) * 1024 + lane * 16);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        out[blockIdx.x * 512 + tid] = s;
    } else {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const char* base = smem + (ks & 1) * 65536;
            asm volatile("" ::: "memory");
            // 32x32x16: A / B operand = 32 rows x 16 k, lane l: row l % 32, k = 8 (l / 32) .. + 7: one 1-KiB block per (32-row block, 16-k step)
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                bf16x8 wf[4], xf[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(base + ((wave_n * 4 + i) * 4 + k4) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) xf[j] = *(const bf16x8*)(base + 32768 + ((wave_m * 2 + j) * 4 + k4) * 1024 + lane * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        out[blockIdx.x * 512 + tid] = s;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const double warm_s = argc > 1 ? atof(argv[1]) : 2.0;
    const int blocks = 256;
    std::vector<uint16_t> h(65536 * 8);
    uint32_t st = 12345u;
    for (auto& v : h) {                     // random bf16 in roughly [-1, 1): random sign, exponent 120..126, random mantissa
        st = st * 1664525u + 1013904223u;
        const uint32_t r = st >> 8;
        v = (uint16_t)(((r & 1) << 15) | ((120 + (r >> 1) % 7) << 7) | ((r >> 8) & 127));
    }
    bf16x8* init; float* out; unsigned long long* stamps;
    CK(hipMalloc(&init, h.size() * 2)); CK(hipMalloc(&out, blocks * 512 * 4)); CK(hipMalloc(&stamps, blocks * 16));
    CK(hipMemcpy(init, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)probe<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void*)probe<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int shape, int reps, double* ms, double* clk_ghz, double* cyc_per_kstep) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
            if (shape == 16) probe<16><<<blocks, 512, 131072>>>(init, out, stamps);
            else probe<32><<<blocks, 512, 131072>>>(init, out, stamps);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        *ms = t / reps;
        std::vector<unsigned long long> hs(blocks * 2);
        CK(hipMemcpy(hs.data(), stamps, blocks * 16, hipMemcpyDeviceToHost));
        std::vector<double> clk, cyc;
        for (int b = 0; b < blocks; ++b) { clk.push_back((double)hs[2 * b] / (double)hs[2 * b + 1] * 0.1); cyc.push_back((double)hs[2 * b] / KSTEPS); }
        std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
        *clk_ghz = clk[blocks / 2]; *cyc_per_kstep = cyc[blocks / 2];
    };
    const double flop = 2.0 * 256 * 256 * 64 * (double)KSTEPS * blocks;
    double ms, clk, cyc;
    // warm-up: back-to-back launches of both shapes until the clock has settled
    int wreps = 0;
    for (double spent = 0; spent < warm_s * 1e3;) { run(16, 4, &ms, &clk, &cyc); spent += 4 * ms; run(32, 4, &ms, &clk, &cyc); spent += 4 * ms; ++wreps; }
    printf("{\"probe\": \"mfma_shape\", \"tile_per_wave\": \"128x64\", \"waves_per_simd\": 2, \"ksteps\": %d, \"rounds\": [", KSTEPS);
    for (int round = 0; round < 6; ++round) {
        for (int shape : {16, 32}) {
            run(shape, 8, &ms, &clk, &cyc);
            printf("%s{\"shape\": \"%s\", \"ms\": %.4f, \"tflops\": %.1f, \"clock_ghz\": %.3f, \"cycles_per_kstep\": %.1f}", (round || shape == 32) ? ", " : "",
                   shape == 16 ? "16x16x32" : "32x32x16", ms, flop / (ms * 1e-3) / 1e12, clk, cyc);
        }
    }
    printf("]}\n");
    return 0;
}
