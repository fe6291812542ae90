// Audio front-end of the BEATs processor on the GPU: Kaldi-compatible log-mel filterbank.
//
// Replaces torchaudio.compliance.kaldi.fbank(waveform * 2**15, num_mel_bins=128, sample_frequency=16000, frame_length=25,
// frame_shift=10) + the (x - 15.41663) / (2 * 6.55582) normalisation + zero padding of
// modelcompose/model/multimodal_encoder/beats/audio_processor.py:143-170 (torchaudio is a third-party CPU dependency of the
// reference).  One workgroup per frame: DC removal, pre-emphasis 0.97, povey window, 512-point radix-2 FFT in LDS, power
// spectrum, 128 triangular mel filters (kaldi mel scale, 20 Hz .. Nyquist), log(max(e, eps)).  Frames past the signal are the
// zero padding the processor appends after normalising.
#include "common.h"

#define FB_WIN 400
#define FB_SHIFT 160
#define FB_FFT 512
#define FB_MEL 128

__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ wav, const int32_t* __restrict__ n_samples, int64_t wav_stride,
                                                    const float* __restrict__ window, const float* __restrict__ mel /*[128][257]*/,
                                                    const int32_t* __restrict__ mel_lo, const int32_t* __restrict__ mel_hi, float in_scale,
                                                    float mean, float inv_2std, bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32,
                                                    int frames_out) {
    __shared__ float re[FB_FFT], im[FB_FFT];
    __shared__ float tw_re[FB_FFT / 2], tw_im[FB_FFT / 2];
    __shared__ float red[16];
    const int frame = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ns = n_samples[b];
    const int n_frames = ns >= FB_WIN ? 1 + (ns - FB_WIN) / FB_SHIFT : 0;
    const int64_t orow = ((int64_t)b * frames_out + frame) * FB_MEL;
    if (frame >= n_frames) {                         // zero padding (audio_processor.py:155-158)
        if (tid < FB_MEL) {
            if (out_bf16) out_bf16[orow + tid] = (bf16_t)0.0f;
            if (out_f32) out_f32[orow + tid] = 0.0f;
        }
        return;
    }
    const float* x = wav + (int64_t)b * wav_stride + (int64_t)frame * FB_SHIFT;
    // twiddles exp(-2 pi i k / 512)
    {
        float s, c;
        sincospif(-2.0f * (float)tid / (float)FB_FFT, &s, &c);
        tw_re[tid] = c; tw_im[tid] = s;
    }
    // DC removal
    float v0 = tid < FB_WIN ? x[tid] * in_scale : 0.f;
    float v1 = tid + 256 < FB_WIN ? x[tid + 256] * in_scale : 0.f;
    const float mu = block_sum(v0 + v1, red) / (float)FB_WIN;
    // pre-emphasis on the DC-free frame with replicate padding on the left, povey window, bit-reversed store
    auto sample = [&](int i) { return x[i] * in_scale - mu; };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int i = tid + half * 256;
        float v = 0.f;
        if (i < FB_WIN) v = (sample(i) - 0.97f * sample(i > 0 ? i - 1 : 0)) * window[i];
        const int r = __brev((unsigned)i) >> (32 - 9);
        re[r] = v; im[r] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < 9; ++st) {
        const int half = 1 << st;
        const int j = tid & (half - 1);
        const int i0 = ((tid >> st) << (st + 1)) + j, i1 = i0 + half;
        const int tw = j << (8 - st);
        const float wr = tw_re[tw], wi = tw_im[tw];
        const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
        const float tr = br * wr - bi * wi, ti = br * wi + bi * wr;
        __syncthreads();
        re[i0] = ar + tr; im[i0] = ai + ti;
        re[i1] = ar - tr; im[i1] = ai - ti;
        __syncthreads();
    }
    // power spectrum of bins 0..256 in re[] (bin 256 lives at index 256)
    {
        const float p0 = re[tid] * re[tid] + im[tid] * im[tid];
        const float p1 = re[tid + 256] * re[tid + 256] + im[tid + 256] * im[tid + 256];
        __syncthreads();
        re[tid] = p0; re[tid + 256] = p1;
        __syncthreads();
    }
    if (tid < FB_MEL) {
        float e = 0.f;
        const float* mrow = mel + tid * (FB_FFT / 2 + 1);
        for (int k = mel_lo[tid]; k <= mel_hi[tid]; ++k) e += re[k] * mrow[k];
        const float lg = logf(fmaxf(e, 1.1920929e-07f));
        const float o = (lg - mean) * inv_2std;
        if (out_bf16) out_bf16[orow + tid] = (bf16_t)o;
        if (out_f32) out_f32[orow + tid] = o;
    }
}

// wav [B, wav_stride] fp32 in [-1, 1] (16 kHz), n_samples [B] valid samples per row (device).  out [B, frames_out, 128]
// (bf16 and / or fp32): frame f = normalised log-mel of samples [160 f, 160 f + 400); frames beyond the signal are zero.
// window [400], mel [128, 257], mel_lo / mel_hi [128] (first / last non-zero column) come from the host (oracle-free tables in
// modelcompose_amd/model/audio_processor.py).  in_scale = 2**15, mean / std = 15.41663 / 6.55582 for BEATs.
extern "C" int mc_fbank_f32(const float* wav, const int32_t* n_samples, int64_t wav_stride, int B, const float* window, const float* mel,
                            const int32_t* mel_lo, const int32_t* mel_hi, float in_scale, float mean, float std, void* out_bf16, float* out_f32,
                            int frames_out, void* stream) {
    MC_CHECK_ARG(wav && n_samples && window && mel && mel_lo && mel_hi && (out_bf16 || out_f32) && B > 0 && frames_out > 0 && std > 0.f,
                 "mc_fbank_f32: bad arguments");
    dim3 grid(frames_out, B);
    fbank_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(wav, n_samples, wav_stride, window, mel, mel_lo, mel_hi, in_scale, mean, 1.0f / (2.0f * std),
                                                        (bf16_t*)out_bf16, out_f32, frames_out);
    MC_CHECK_LAUNCH();
    return 0;
}
