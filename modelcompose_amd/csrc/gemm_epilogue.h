// Epilogue description and the 4-column store shared by every GEMM kernel of the library (gemm.hip: the tile kernels; gemm_strip.hip: the
// M <= 64 strip kernel).
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------
// epilogue shared by both kernels: lane owns out[m][n..n+3]
// ------------------------------------------------------------------------------------------
struct Epilogue {
    const bf16_t* bias;      // [N] or null
    const bf16_t* residual;  // [M, ldr] or null
    int64_t ldr;
    void* out;
    int64_t ldo;
    int act;
    int out_f32;
    float alpha;             // scale applied to the accumulator before bias
    float beta;              // scale applied to the residual
    const float* row_scale;  // [M] fp32 or null: accumulator row m is multiplied by row_scale[m] (RMSNorm 1/rms with the
                             // norm weight folded into W at compose time: LlamaRMSNorm, multimodal_llama.py:405-406)
    int swiglu;              // 16-row weight blocks alternate gate/up: out[m][16*(nb/2) + c] = silu(gate) * up (LocalLoraMLP :381-388)
    float rms_eps;           // > 0 (strip kernel only): row m is scaled by rsqrt(mean_k x[m][k]^2 + rms_eps), computed from the x fragments the
                             // kernel streams anyway (LlamaRMSNorm factor without a separate pass; replaces row_scale)
    // q_out != null (256x256 kernel, 256-column tiles, D = 128): the launch is a prefill's q|k|v projection; a wave's 128 columns are one
    // head, whose halves d / d + 64 sit in the same lane (acc[0][i] / acc[1][i]): rotate in registers and scatter to q_out / the caches
    struct Rope {
        const int32_t* row_b; const int32_t* row_pos; const int32_t* row_t;
        const float* cosT; const float* sinT;
        bf16_t* q_out; bf16_t* k_cache; bf16_t* v_cache;
        int H, Hkv, Lq, Smax;
    } rope;
    float* ss_parts;         // non-null (256x256 kernel, wide plain epilogue): ss_parts[m * ss_chunks + n / 128] = sum of squares of the stored
    int ss_chunks;           // bf16 values of row m in that 128-column chunk (mc_gemm_args.rms_out)
};

__device__ __forceinline__ void epilogue_store4(const Epilogue& e, int m, int n, f32x4 v) {
    const float a = e.row_scale ? e.alpha * e.row_scale[m] : e.alpha;
    float r[4] = {v[0] * a, v[1] * a, v[2] * a, v[3] * a};
    if (e.bias) {
        bf16x4 b = *(const bf16x4*)(e.bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] += (float)b[i];
    }
    if (e.act != MC_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = mc_act(r[i], e.act);
    }
    if (e.residual) {
        bf16x4 b = *(const bf16x4*)(e.residual + (int64_t)m * e.ldr + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] += e.beta * (float)b[i];
    }
    if (e.out_f32) {
        f32x4 o = {r[0], r[1], r[2], r[3]};
        *(f32x4*)((float*)e.out + (int64_t)m * e.ldo + n) = o;
    } else {
        bf16x4 o = {(bf16_t)r[0], (bf16_t)r[1], (bf16_t)r[2], (bf16_t)r[3]};
        *(bf16x4*)((bf16_t*)e.out + (int64_t)m * e.ldo + n) = o;
    }
}

// gate / up accumulators of the same 4 output columns (n_out = column in the [M, N/2] result)
__device__ __forceinline__ void epilogue_store4_swiglu(const Epilogue& e, int m, int n_out, f32x4 g, f32x4 u) {
    const float a = e.row_scale ? e.alpha * e.row_scale[m] : e.alpha;
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // gate and up are rounded to the storage dtype as the unfused path stores them (the reference's projections return
        // the model dtype, :381-388); act(gate) * up is evaluated in fp32 and rounded once
        const float gg = (float)(bf16_t)(g[i] * a), uu = (float)(bf16_t)(u[i] * a);
        o[i] = (bf16_t)(gg * mc_sigmoid(gg) * uu);
    }
    *(bf16x4*)((bf16_t*)e.out + (int64_t)m * e.ldo + n_out) = o;
}


// gemm_strip.hip: every launch of at most 64 rows (one K-reduction order per (N, K), whatever M is - see the kernel's comment)
int mc_strip_launch(const bf16_t* x, int64_t ldx, const bf16_t* w_packed, int M, int N, int K, const Epilogue& ep, hipStream_t s);
