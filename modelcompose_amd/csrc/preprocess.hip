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

// ------------------------------------------------------------------------------------------
// Image front-end of the vision branch: expand2square (modelcompose/mm_utils.py:14-26) + the CLIPImageProcessor steps
// (resize with PIL's 8-bit bicubic resampling, centre crop, rescale 1/255, normalise) that the reference runs on the CPU through
// PIL / transformers.  Integer work restated exactly: 22-bit fixed-point weights (computed on the host in double precision, as
// Pillow's precompute_coeffs / normalize_coeffs_8bpc do), rounding bias, clamp to 0..255 after EACH pass (horizontal, then vertical).
#define RS_PRECISION_BITS 22

struct ResampleSrc {
    const uint8_t* img;       // [h, w, 3] uint8 (HWC)
    int h, w;                 // real image
    int off_y, off_x;         // paste offset inside the virtual (padded) canvas
    int bg[3];                // canvas colour outside the image
};

__device__ __forceinline__ int rs_get(const ResampleSrc& s, int y, int x, int c) {
    const int yy = y - s.off_y, xx = x - s.off_x;
    if (yy < 0 || yy >= s.h || xx < 0 || xx >= s.w) return s.bg[c];
    return s.img[((int64_t)yy * s.w + xx) * 3 + c];
}

__device__ __forceinline__ int rs_clip8(int acc) {
    const int v = acc >> RS_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass over canvas rows [row0, row0 + rows): tmp[r][xx][c]; with_resize == 0 copies the canvas column xx
__global__ __launch_bounds__(256) void resample_h_kernel(ResampleSrc s, const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                         int row0, int rows, int out_w, int with_resize, uint8_t* __restrict__ tmp) {
    const int64_t total = (int64_t)rows * out_w;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / out_w), xx = (int)(i % out_w);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int v;
            if (with_resize) {
                const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
                int acc = 1 << (RS_PRECISION_BITS - 1);
                for (int k = 0; k < n; ++k) acc += kk[xx * ksize + k] * rs_get(s, row0 + r, x0 + k, c);
                v = rs_clip8(acc);
            } else {
                v = rs_get(s, row0 + r, xx, c);
            }
            tmp[i * 3 + c] = (uint8_t)v;
        }
    }
}

// vertical pass over the crop window + rescale + normalise: out[c][y][x] (CHW), tmp rows are relative to row0
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, int tmp_w, int row0, const int32_t* __restrict__ bounds,
                                                              const int32_t* __restrict__ kk, int ksize, int with_resize, int top, int left, int size_h,
                                                              int size_w, float m0, float m1, float m2, float s0, float s1, float s2,
                                                              bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32, uint8_t* __restrict__ out_u8) {
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    const int64_t total = (int64_t)size_h * size_w;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i / size_w), x = (int)(i % size_w);
        const int yy = top + y, xx = left + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int v;
            if (with_resize) {
                const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
                int acc = 1 << (RS_PRECISION_BITS - 1);
                for (int k = 0; k < n; ++k) acc += kk[yy * ksize + k] * (int)tmp[((int64_t)(y0 + k - row0) * tmp_w + xx) * 3 + c];
                v = rs_clip8(acc);
            } else {
                v = tmp[((int64_t)(yy - row0) * tmp_w + xx) * 3 + c];
            }
            if (out_u8) out_u8[i * 3 + c] = (uint8_t)v;
            // transformers: float32(float64(v) * (1 / 255)), then (x - mean) / std in float32
            const float f = (float)((double)v * (1.0 / 255.0));
            const float o = (f - mean[c]) / stdv[c];
            const int64_t oi = (int64_t)c * total + i;
            if (out_bf16) out_bf16[oi] = (bf16_t)o;
            if (out_f32) out_f32[oi] = o;
        }
    }
}

// img [h, w, 3] uint8 on the device is pasted at (off_y, off_x) of a canvas_h x canvas_w canvas of colour bg (expand2square; pass the
// image size and zero offsets for no padding), resized to res_h x res_w with PIL's bicubic (coefficient tables from the host:
// bounds_* [out, 2], kk_* [out, ksize_*]; a null table = that axis keeps its size), centre-cropped at (top, left) to size_h x size_w,
// rescaled and normalised to out [3, size_h, size_w] (bf16 and / or fp32; out_u8 optionally receives the cropped uint8 HWC image).
// tmp: scratch of canvas_h * res_w * 3 bytes.
extern "C" int mc_image_preprocess_u8(const void* img, int h, int w, int canvas_h, int canvas_w, int off_y, int off_x, const int32_t* bg,
                                      const int32_t* bounds_h, const int32_t* kk_h, int ksize_h, const int32_t* bounds_v, const int32_t* kk_v,
                                      int ksize_v, int res_h, int res_w, int top, int left, int size_h, int size_w, const float* mean,
                                      const float* stdv, void* tmp, void* out_bf16, float* out_f32, void* out_u8, void* stream) {
    MC_CHECK_ARG(img && bg && mean && stdv && tmp && (out_bf16 || out_f32 || out_u8) && h > 0 && w > 0, "mc_image_preprocess_u8: null / empty argument");
    MC_CHECK_ARG(canvas_h >= h && canvas_w >= w && off_y >= 0 && off_x >= 0 && off_y + h <= canvas_h && off_x + w <= canvas_w,
                 "mc_image_preprocess_u8: image does not fit the canvas");
    MC_CHECK_ARG((bounds_h ? res_w > 0 : res_w == canvas_w) && (bounds_v ? res_h > 0 : res_h == canvas_h), "mc_image_preprocess_u8: bad resize");
    MC_CHECK_ARG(top >= 0 && left >= 0 && top + size_h <= res_h && left + size_w <= res_w && size_h > 0 && size_w > 0, "mc_image_preprocess_u8: bad crop");
    hipStream_t s = (hipStream_t)stream;
    ResampleSrc src{(const uint8_t*)img, h, w, off_y, off_x, {bg[0], bg[1], bg[2]}};
    const int64_t t1 = (int64_t)canvas_h * res_w;
    resample_h_kernel<<<(int)min((int64_t)4096, (t1 + 255) / 256), 256, 0, s>>>(src, bounds_h, kk_h, ksize_h, 0, canvas_h, res_w, bounds_h != nullptr,
                                                                                (uint8_t*)tmp);
    const int64_t t2 = (int64_t)size_h * size_w;
    resample_v_norm_kernel<<<(int)min((int64_t)4096, (t2 + 255) / 256), 256, 0, s>>>((const uint8_t*)tmp, res_w, 0, bounds_v, kk_v, ksize_v,
                                                                                     bounds_v != nullptr, top, left, size_h, size_w, mean[0], mean[1],
                                                                                     mean[2], stdv[0], stdv[1], stdv[2], (bf16_t*)out_bf16, out_f32,
                                                                                     (uint8_t*)out_u8);
    MC_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// LanguageBind video front-end after decoding (languagebind/video/processing_video.py:24-68): frames [T, H, W, 3] uint8 ->
// x / 255 -> (x - mean) / std -> bilinear resize (align_corners = False, no antialiasing: torch.nn.functional.interpolate as called by
// pytorchvideo's ShortSideScale) -> centre crop -> optional horizontal flip -> out [3, T, size, size].  Source index and
// interpolation formula follow ATen's upsample_bilinear2d (area_pixel_compute_source_index, h0*(w0*p00 + w1*p01) + h1*(...)).
__global__ __launch_bounds__(256) void video_preprocess_kernel(const uint8_t* __restrict__ frames, int T, int H, int W, int res_h, int res_w, int top,
                                                               int left, int size, int flip, float m0, float m1, float m2, float s0, float s1,
                                                               float s2, bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32) {
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    const float sh = (float)H / (float)res_h, sw = (float)W / (float)res_w;
    const int64_t per_frame = (int64_t)size * size, total = (int64_t)T * per_frame;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i / per_frame);
        const int y = (int)((i % per_frame) / size), x = (int)(i % size);
        const int oy = top + y, ox = left + (flip ? size - 1 - x : x);
        const float fy = fmaxf(sh * ((float)oy + 0.5f) - 0.5f, 0.f), fx = fmaxf(sw * ((float)ox + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float h1 = fy - (float)y0, h0 = 1.f - h1, w1 = fx - (float)x0, w0 = 1.f - w1;
        const uint8_t* f = frames + (int64_t)t * H * W * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            auto px = [&](int yy, int xx) { return ((float)f[((int64_t)yy * W + xx) * 3 + c] / 255.0f - mean[c]) / stdv[c]; };
            const float v = h0 * (w0 * px(y0, x0) + w1 * px(y0, x1)) + h1 * (w0 * px(y1, x0) + w1 * px(y1, x1));
            const int64_t oi = ((int64_t)c * T + t) * per_frame + (int64_t)y * size + x;
            if (out_bf16) out_bf16[oi] = (bf16_t)v;
            if (out_f32) out_f32[oi] = v;
        }
    }
}

extern "C" int mc_video_preprocess_u8(const void* frames, int T, int H, int W, int res_h, int res_w, int top, int left, int size, int flip,
                                      const float* mean, const float* stdv, void* out_bf16, float* out_f32, void* stream) {
    MC_CHECK_ARG(frames && mean && stdv && (out_bf16 || out_f32) && T > 0 && H > 0 && W > 0 && res_h >= size && res_w >= size && size > 0,
                 "mc_video_preprocess_u8: bad arguments");
    MC_CHECK_ARG(top >= 0 && left >= 0 && top + size <= res_h && left + size <= res_w, "mc_video_preprocess_u8: bad crop");
    const int64_t total = (int64_t)T * size * size;
    video_preprocess_kernel<<<(int)min((int64_t)8192, (total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        (const uint8_t*)frames, T, H, W, res_h, res_w, top, left, size, flip, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2],
        (bf16_t*)out_bf16, out_f32);
    MC_CHECK_LAUNCH();
    return 0;
}
