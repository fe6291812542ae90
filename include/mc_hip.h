/* mc_hip.h — C ABI of libmc_hip.so, the MI355X (gfx950) forward/generation path for ModelCompose.
 *
 * Boundary rules: plain pointers and sizes only, no torch types.  Every pointer is a
 * BORROWED device pointer valid for the duration of the call (the caller — PyTorch-ROCm in the shipped
 * host layer — owns all memory); `stream` is a hipStream_t (pass torch's current HIP stream).  Launches
 * are asynchronous on that stream and graph-capturable (no allocation / synchronisation inside).
 * Return value: 0 = ok, 1 = invalid argument, 2 = HIP error; mc_last_error() returns the thread-local
 * message.  The Python wrapper raises ValueError for 1 (the reference raises ValueError on shape errors,
 * multimodal_llama.py:297-318) and RuntimeError for 2.
 *
 * The reference has no FFI seam (it is pure Python on third-party CUDA wheels); each entry point below
 * names the reference computation it replaces (paths relative to /root/reference/modelcompose).
 * dtype: bf16 storage, fp32 accumulation.
 */
#ifndef MC_HIP_H
#define MC_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MC_ABI_VERSION 10

/* activation codes for mc_gemm_bf16 */
#define MC_ACT_NONE 0
#define MC_ACT_GELU 1        /* exact erf GELU: multimodal_projector/builder.py:208-215 (nn.GELU)          */
#define MC_ACT_QUICK_GELU 2  /* x*sigmoid(1.702x): CLIP MLP (transformers CLIPMLP via clip_encoder.py:53) */
#define MC_ACT_SILU 3        /* ACT2FN['silu']: model/language_model/multimodal_llama.py:361               */
#define MC_ACT_RELU 4        /* nn.ReLU of the PointBERT mini-PointNet (multimodal_encoder/pointbert/dvae.py:196-206) */

const char* mc_last_error(void);
int mc_abi_version(void);
int mc_device_info(int* cu_count, int64_t* hbm_bytes, char* arch, int arch_len);
/* The 16-bit storage element of this build: MC_DTYPE_BF16 (libmc_hip.so - BASELINE.json's dtype, the headline) or MC_DTYPE_F16
 * (libmc_hip_f16.so: the SAME sources and entry points instantiated on IEEE half, fp32 accumulation unchanged - the reference's own
 * inference dtype, model/builder.py:41, :162, :185, kept as the parity instrument: 8x finer mantissa at the same MFMA rate).  Every
 * `_bf16` entry point below takes / returns that element type in the f16 build; fp32 and integer interfaces are unchanged. */
int mc_storage_dtype(void);

/* ---- weights ---------------------------------------------------------------------------------------
 * Packed layout (see csrc/gemm.hip): [ceil16(N)/16][ceil64(K)/32][64 lanes][8] bf16, zero padded.        */
int mc_packed_weight_elems(int N, int K, int64_t* out_elems);
int mc_pack_weight_bf16(const void* w_rowmajor, int64_t ldw, void* packed, int N, int K, void* stream);
/* same, reading w[n * stride_n + k * stride_k] (stride_n = 1, stride_k = ld packs the transpose of a row-major [K, N] matrix) */
int mc_pack_weight_strided_bf16(const void* w, int64_t stride_n, int64_t stride_k, void* packed, int N, int K, void* stream);
/* n packs in one launch; descs_dev: DEVICE array of { const void* src; void* dst; int64_t stride_n, stride_k; int32_t N, K; } (32 bytes each);
 * every descriptor is processed by blocks_per_desc workgroups (the finetune step refreshes all LoRA fragment images this way) */
int mc_pack_weight_batch_bf16(const void* descs_dev, int n, int blocks_per_desc, void* stream);
int mc_unpack_weight_bf16(const void* packed, void* w_rowmajor, int N, int K, void* stream);

/* W' = W + sum_i scale[i] * B_i * A_i  -> packed (and optionally row-major).  Replaces the per-forward
 * composition of LocalLoraLinear.forward (model/language_model/multimodal_llama.py:130-157) and restates
 * scripts/evaluate_delta_weights.py:8-15.  at_list[i] = A_i^T [K, r]; b_list[i] = B_i [N, r]; r % 32 == 0. */
int mc_compose_weight_bf16(const void* w_rowmajor, int64_t ldw, const void* const* at_list, const void* const* b_list,
                           const float* scales, int n_terms, int r, void* out_packed, void* out_rowmajor, int64_t ldo,
                           int N, int K, void* stream);

/* As above with (a) col_scale fp32 [K] multiplied into the columns before the single bf16 rounding — folds the weight of the
 * LlamaRMSNorm that precedes the linear (multimodal_llama.py:405-406, :443, :462) into it — and (b) block interleaving: packed
 * 16-row block nb is written at block index nb*nb_stride + nb_offset (gate_proj / up_proj interleaved for the fused SwiGLU
 * epilogue of mc_gemm_ex_bf16).  (c) retention_parts (optional, device, mc_compose_retention_floats(N, K) floats): per-workgroup
 * partial sums {sum (W' - bf16(W c)) (dW c), sum (dW c)^2} over the elements the workgroup wrote - their ratio, summed by the caller,
 * is the share of the delta that survives the single bf16 rounding (1 for trained deltas; falls when |dW| is below half a bf16 step
 * of W: the reference's branch form, multimodal_llama.py:130-149, has no such rounding).                                        */
int mc_compose_weight_ex_bf16(const void* w_rowmajor, int64_t ldw, const void* const* at_list, const void* const* b_list,
                              const float* scales, int n_terms, int r, void* out_packed, void* out_rowmajor, int64_t ldo,
                              int N, int K, const float* col_scale, int nb_stride, int nb_offset, float* retention_parts, void* stream);

/* As mc_compose_weight_ex_bf16 with dither_seed != 0: the single bf16 rounding is UNBIASED - the magnitude is rounded up with probability
 * equal to the discarded fraction, the uniform bits from a counter hash of (dither_seed, n, k), reproducible - so that E[W'] is the fp32
 * composition.  For adapters whose delta lies below half a bf16 step of W (retention < 0.9 under round-to-nearest: W + dW rounds back
 * to W, a SYSTEMATIC loss the reference's branch form, multimodal_llama.py:130-149, does not have) the delta is kept in expectation, at
 * the price of rounding noise of the size an off-grid weight always has (<= 1 bf16 step, zero mean).  dither_seed 0 = the ex form.  */
int mc_compose_weight_dither_bf16(const void* w_rowmajor, int64_t ldw, const void* const* at_list, const void* const* b_list,
                                  const float* scales, int n_terms, int r, void* out_packed, void* out_rowmajor, int64_t ldo,
                                  int N, int K, const float* col_scale, int nb_stride, int nb_offset, float* retention_parts,
                                  uint32_t dither_seed, void* stream);

/* ONE pass per linear for ALL routed adapters (round 5): the W tile is read once and every output o = 0 .. n_out - 1 is written from it,
 * W'_o = bf16((W + sum over the terms m with bit m of term_mask[o] set of scale[m] B_m A_m) diag(col_scale)) - 1 read + n_out writes of W
 * instead of n_out x (read + write).  For the 3-way composed model: terms = {default-vision, default-audio, default-video, vision, audio,
 * video}, outputs = {default: the first three, vision, audio, video: one each} (LocalLoraLinear.forward, multimodal_llama.py:130-157).
 * term_mask NULL = every term in every output; dither_seeds / retention_parts / out_rowmajor may be NULL or hold NULL / 0 entries;
 * retention_parts[o] holds mc_compose_retention_floats(N, K) floats ({num, den} per workgroup, summed by the caller).  At most 8 terms,
 * 6 outputs.  mc_compose_weight_{,ex_,dither_}bf16 are this call with one output. */
typedef struct mc_compose_multi_args {
    const void* w; int64_t ldw;
    const void* const* at_list; const void* const* b_list; const float* scales; int n_terms, r;
    int n_out; void* const* out_packed; void* const* out_rowmajor; const uint32_t* term_mask; const uint32_t* dither_seeds;
    float* const* retention_parts;
    int64_t ldo; int N, K; const float* col_scale; int nb_stride, nb_offset;
} mc_compose_multi_args;
int mc_compose_multi_bf16(const mc_compose_multi_args* args, void* stream);
/* The same for n linears in ONE launch (round 6: a model's 224 linears were 224 launches of 180-530 us): args[0 .. n - 1] (host array).
 * Linears with K % 64 == 0, r % 64 == 0 and 16-byte aligned rows share one launch of the LDS-staged tile kernel (128 x 256 tiles, the LoRA
 * factors through a 3-stage LDS-DMA ring, W read once per tile into registers); any other linear takes the general kernel, one launch
 * each.  Results are bit-identical either way (retention partials included).  The descriptor table is copied to a library-owned device
 * buffer on `stream` (the call synchronises that stream once): a load-time call, not for graph capture. */
int mc_compose_batch_bf16(const mc_compose_multi_args* args, int n, void* stream);
int mc_compose_retention_floats(int N, int K, int64_t* floats);

/* ---- audio front-end: Kaldi log-mel filterbank + BEATs normalisation + zero padding (beats/audio_processor.py:143-170; replaces
 * torchaudio.compliance.kaldi.fbank, a third-party CPU dependency of the reference).  wav [B, wav_stride] fp32 at 16 kHz,
 * n_samples [B] (device), out [B, frames_out, 128] bf16 and / or fp32; window [400], mel [128, 257], mel_lo / mel_hi [128].    */
int mc_fbank_f32(const float* wav, const int32_t* n_samples, int64_t wav_stride, int B, const float* window, const float* mel,
                 const int32_t* mel_lo, const int32_t* mel_hi, float in_scale, float mean, float std, void* out_bf16, float* out_f32,
                 int frames_out, void* stream);

/* ---- image front-end: expand2square (mm_utils.py:14-26) + CLIPImageProcessor (PIL 8-bit bicubic resize, centre crop, /255,
 * normalise) as integer-exact device kernels.  img [h, w, 3] uint8 (device) pasted at (off_y, off_x) of a canvas of colour bg
 * (host int[3]); bounds / kk = Pillow's precompute_coeffs tables in 22-bit fixed point (device int32; NULL = axis not resized);
 * mean / stdv host float[3]; tmp = canvas_h * res_w * 3 bytes of scratch; out [3, size_h, size_w].                          */
int mc_image_preprocess_u8(const void* img, int h, int w, int canvas_h, int canvas_w, int off_y, int off_x, const int32_t* bg,
                           const int32_t* bounds_h, const int32_t* kk_h, int ksize_h, const int32_t* bounds_v, const int32_t* kk_v,
                           int ksize_v, int res_h, int res_w, int top, int left, int size_h, int size_w, const float* mean,
                           const float* stdv, void* tmp, void* out_bf16, float* out_f32, void* out_u8, void* stream);

/* ---- video front-end after decoding (languagebind/video/processing_video.py:24-68): frames [T, H, W, 3] uint8 (device) -> /255 ->
 * normalise -> bilinear resize to res_h x res_w (torch interpolate semantics, align_corners False) -> crop (top, left, size) ->
 * optional horizontal flip -> out [3, T, size, size] bf16 and / or fp32.  mean / stdv: host float[3].                        */
int mc_video_preprocess_u8(const void* frames, int T, int H, int W, int res_h, int res_w, int top, int left, int size, int flip,
                           const float* mean, const float* stdv, void* out_bf16, float* out_f32, void* stream);

/* ---- TIES merging of checkpoints (scripts/model_composition/ties_merging.py:88-221, --strategy ties-{mean,sum,max}) ------------
 * x: n flattened task vectors [n, d] (row stride ld elements) of dtype MC_DTYPE_*.  mc_ties_hist is one pass of the exact radix
 * select of each row's k-th smallest magnitude (host reads the 2048-bin histograms and picks the bin: 3 passes of 11/11/10 bits);
 * mc_ties_merge elects the signs (zero sums take the majority sign) and aggregates the agreeing trimmed entries.                 */
#define MC_DTYPE_F32 0
#define MC_DTYPE_BF16 1
#define MC_DTYPE_F16 2
int mc_ties_hist(const void* x, int dtype, int64_t ld, int64_t d, int n, int shift, int nbins, const uint32_t* prefix, int prefix_shift,
                 uint32_t* hist, void* stream);
int mc_ties_merge(const void* x, int dtype, int64_t ld, int64_t d, int n, const float* thr, int8_t* sign, long long* sign_sum, int func,
                  void* out, void* stream);

/* ---- parameter-interference metrics (scripts/model_composition/calculate_metrics.py:26-37, :58-67) ----------------------------
 * One pass over the n >= 2 task vectors x [n, d]; every workgroup writes 8 double partial sums to partial [mc_merge_metrics_blocks()][8]:
 * sum (x0-x1)^2, sum x0 x1, sum x0^2, sum x1^2 (L2 / cosine of rows 0, 1), then sum_j |sum_i x_ij| / sum_i |x_ij| and the count of
 * columns with a non-zero magnitude sum (SSD), and the same pair with each row trimmed to |x| >= thr[i] (TSSD; thr may be NULL).  */
int mc_merge_metrics(const void* x, int dtype, int64_t ld, int64_t d, int n, const float* thr, double* partial, void* stream);
int mc_merge_metrics_blocks(void);

/* ---- linear: out[M,N] = act(alpha * x[M,K] W^T + bias) + beta * residual ----------------------------
 * Replaces F.linear at multimodal_llama.py:122 (LocalLoRA base GEMM), :720 (lm_head), the CLIP / projector
 * linears.  K must be a multiple of 64 (zero padded), x rows 16-byte aligned.  out_f32 != 0 -> fp32 out.  */
int mc_gemm_bf16(const void* x, int64_t ldx, const void* w_packed, const void* bias, const void* residual, int64_t ldr,
                 void* out, int64_t ldo, int M, int N, int K, int act, int out_f32, float alpha, float beta, void* stream);

/* Extended form.  row_scale: fp32 [M] multiplied into accumulator row m before bias/act (the 1/rms factor of LlamaRMSNorm,
 * multimodal_llama.py:405-406, when the norm weight has been folded into W by mc_compose_weight_ex_bf16).
 * swiglu: the packed weight interleaves gate_proj / up_proj per 16-row block (block 2j = gate rows 16j.., block 2j+1 = up rows
 * 16j..); out is [M, N/2] bf16 = silu(gate) * up (LocalLoraMLP.forward, multimodal_llama.py:381-388).
 * rms_eps > 0 (strip family, no row_scale): the kernel computes the RMSNorm factor rsqrt(mean_k x[m][k]^2 + rms_eps) of every row from the
 * x fragments it streams anyway and uses it as row_scale - the decode path needs no separate normalisation pass.
 * split_k: 1, or < 0 (tile family) = "auto" - launches that would leave most CUs idle (few output tiles, long K: the LoRA rank projections
 * of the finetune step) are split along K into library-owned fp32 slabs, summed in fixed order, then given the normal epilogue.     */
typedef struct mc_gemm_args {
    const void* x; int64_t ldx; const void* w_packed; const void* bias; const void* residual; int64_t ldr;
    void* out; int64_t ldo; int M, N, K; int act; int out_f32; float alpha, beta;
    const float* row_scale; int swiglu; int split_k; float rms_eps;
    const struct mc_rope_scatter* rope;   /* non-null: the launch is the q|k|v projection of a prefill and its epilogue does what mc_rope_kv_bf16
                                           * would do next (below); `out` is then scratch ([M, ldo], may or may not be written) */
    float* rms_out; float rms_out_eps;    /* rms_out non-null (bf16 output, no SwiGLU): after the launch rms_out[m] = rsqrt(mean_n out[m][n]^2 +
                                           * rms_out_eps) of the STORED rows - the factor of the RMSNorm that reads the new hidden state
                                           * (LlamaRMSNorm, multimodal_llama.py:405-406), without the separate mc_rms_scale_bf16 pass: the
                                           * 256x256 kernel's epilogue leaves one sum of squares per row and 128-column chunk, a small
                                           * launch adds them in column order; other routes run mc_rms_scale_bf16 after the GEMM */
    int family;                           /* which kernel family runs the launch (round 6).  The two families add the products of a row in
                                           * different fp32 orders, each in ONE order whatever M is, so a caller that needs a row's bits not
                                           * to depend on how many other rows share its launch (the eval loader batches a question
                                           * differently at 1 and at 8 GPUs) pins the family: MC_GEMM_AUTO (0) = strip kernel for M <= 64,
                                           * tile kernels above; MC_GEMM_STRIP (1) = the M <= 64 strip kernel always, larger M as slices of
                                           * 64 rows (decode steps, lm_head, the last-token tail of a prefill); MC_GEMM_TILE (2) = the
                                           * 128x128 / 256x256 tile kernels always, which are bit-identical to each other (prefill, encoders) */
} mc_gemm_args;
enum { MC_GEMM_AUTO = 0, MC_GEMM_STRIP = 1, MC_GEMM_TILE = 2 };
/* RoPE + scatter fused into the q|k|v projection (LlamaAttention.forward, multimodal_llama.py:281-312: rotate q and k, append k / v to the
 * cache): output row r (absolute row index of the launch, as in mc_rope_kv_bf16) belongs to sequence row_b[r] (< 0: padding, skipped), is
 * query row_t[r] of this call and sits at cache position row_pos[r].  N must be (H + 2 Hkv) D.  With D = 128, an even head count and a
 * launch large enough for the 256x256 kernel, the rotation happens on the accumulators' bf16 values in registers and q / K / V go
 * straight to q_out [(b Lq + t)][H D] and the caches [B][Hkv][Smax][D]; otherwise the GEMM writes `out` and mc_rope_kv_bf16 runs after it.
 * Both routes round identically. */
typedef struct mc_rope_scatter {
    const int32_t* row_b; const int32_t* row_pos; const int32_t* row_t;
    const float* cos_table; const float* sin_table;      /* [max_pos][D / 2] fp32 */
    void* q_out; void* k_cache; void* v_cache;
    int H, Hkv, D, Lq, Smax;
} mc_rope_scatter;
int mc_gemm_ex_bf16(const mc_gemm_args* args, void* stream);

/* The same linear over rows grouped by routed adapter (multimodal_llama.py:262-268 made dense): rows
 * [group_start[g], group_start[g+1]) use w_packed[g]; x / out / residual / row_scale are indexed by absolute row.  group_start
 * (n_groups+1) and w_packed (n_groups) are HOST arrays; args->w_packed and args->M are ignored.  One launch when the problem
 * fills the chip with 256x256 tiles, one mc_gemm_ex_bf16 per group otherwise.                                            */
int mc_gemm_grouped_bf16(const mc_gemm_args* args, int n_groups, const int32_t* group_start, const void* const* w_packed, void* stream);

/* live HIP-event timing of the large-M GEMM kernel on its launch stream (bench.py roofline) */
/* library-wide GEMM policy: "tile192" (default 1) lets the large-M kernel use 192-column tiles when they fill the 256 CUs better than
 * 256-column ones (under-filled launches); 0 = callers that run other work beside such launches keep the wider tile.  Does not change
 * results.  (Kernel A/B variants, forced tile shapes and the raster parameters live in the probes build only: csrc/Makefile `probes`.) */
int mc_gemm_set_option(const char* name, int value);
/* Scratch of the tile kernels' rms_out route (16 x 48 MiB, one slot per launching stream so that concurrent streams share nothing),
 * allocated at the first use made outside stream capture.  Call reserve once before capturing when the process may capture before it
 * has launched eagerly (mc_llm_create does); release before destroying a stream that launched GEMMs (gives its slot back). */
int mc_gemm_reserve_workspace(void* stream);
int mc_gemm_release_workspace(void* stream);
int mc_gemm_profile_enable(int on);
int mc_gemm_profile_read(double* total_ms, double* total_flops, int64_t* launches);
int mc_gemm_profile_read_range(int k_min, int k_max, double* total_ms, double* total_flops, int64_t* launches);   /* launches with k_min <= K <= k_max */
int mc_gemm_profile_read_bytes(double* total_bytes);   /* algorithmic HBM bytes (operands read once, output written once) of the same launches */

/* ---- norms: LlamaRMSNorm (multimodal_llama.py:405-406, :482) / nn.LayerNorm (CLIP blocks) -------------- */
int mc_rmsnorm_bf16(const void* x, int64_t ldx, const void* w, void* out, int64_t ldo, int M, int D, float eps, void* stream);
int mc_layernorm_bf16(const void* x, int64_t ldx, const void* w, const void* b, void* out, int64_t ldo, int M, int D,
                      float eps, void* stream);
/* sum_out[r] = bf16(x[r] + table[idx ? idx[r] : r]); out[r] = LayerNorm(sum_out[r]) - one pass for `hidden_states + temporal_embedding`
 * followed by temporal_layer_norm1 (languagebind/video/modeling_video.py:105-115); bit-identical to mc_add_rows_bf16 + mc_layernorm_bf16. */
int mc_add_layernorm_bf16(const void* x, int64_t ldx, const void* table, int64_t ldt, const int32_t* idx, void* sum_out, int64_t lds,
                          const void* w, const void* b, void* out, int64_t ldo, int M, int D, float eps, void* stream);
/* RMSNorm's per-row factor only (its weight is folded into the next linear): row_scale[m] = rsqrt(mean(x[m]^2) + eps)        */
int mc_rms_scale_bf16(const void* x, int64_t ldx, float* row_scale, int M, int D, float eps, void* stream);

/* ---- RoPE + KV-cache append (multimodal_llama.py:281-289; apply_rotary_pos_emb of transformers 4.31) ----
 * qkv rows are in routed order; row_b / row_pos / row_t give batch entry, absolute position and query index. */
int mc_rope_kv_bf16(const void* qkv, int64_t ld, const int32_t* row_b, const int32_t* row_pos, const int32_t* row_t,
                    const float* cos_table, const float* sin_table, void* q_out, void* k_cache, void* v_cache, int M,
                    int H, int Hkv, int D, int Lq, int Smax, void* stream);

/* ---- training step (BASELINE config 5): element-wise / reduction kernels replacing autograd through the decoder layer
 * (multimodal_llama.py:408-468), the shifted CrossEntropyLoss (:722-733) and torch.optim.AdamW ---------------------------- */
int mc_transpose_bf16(const void* in, int64_t ldi, void* out, int64_t ldo, int R, int C, int Rp, void* stream);   /* out[c][r], cols R..Rp-1 zero */
int mc_lora_mask_rows_bf16(void* t, int64_t ld, const int32_t* row_adapter, int M, int r, int n_adapters, int n_cols, void* stream);
/* Weight-gradient GEMM, "TN" form: out[p][q] = alpha * sum_m a[m][p] * b[m][q] (fp32), a [M, P] / b [M, Q] row-major bf16 with row strides
 * lda / ldb (multiples of 8, 16-byte aligned bases).  The autograd products dW = dy^T x of the trainable LoRA / projector linears
 * (train_multimodal.py's backward) without transposing activations in HBM.  n_problems (1..3) same-shape products per launch; workspace =
 * mc_gemm_tn_workspace_floats() floats (0 = not needed).  Deterministic (fixed-order slab reduction, no atomics). */
int mc_gemm_tn_workspace_floats(int M, int P, int Q, int n_problems, int64_t* floats);
int mc_gemm_tn_bf16(const void* const* a, int64_t lda, const void* const* b, int64_t ldb, float* const* out, int64_t ldo, int n_problems,
                    int M, int P, int Q, float alpha, float* workspace, void* stream);
/* nn.Dropout(p) on the LoRA input (multimodal_llama.py:133-148) with a counter-based Philox4x32-10 mask keyed by (seed, stream_id, element):
 * out = (accumulate ? out : 0) + alpha * x * keep / (1 - p).  The same call regenerates the mask in the backward pass.                */
int mc_dropout_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, int M, int K, float p, unsigned long long seed,
                    unsigned int stream_id, int accumulate, float alpha, void* stream);
int mc_rmsnorm_bwd_bf16(const void* x, int64_t ldx, const void* g, const void* dy, int64_t ldy, const void* dres, int64_t ldr,
                        void* dx, int64_t ldd, int M, int D, float eps, void* stream);
/* LayerNorm backward (Q-Former projector, multimodal_projector/Qformer.py:112-130): dx, and t = dy * xhat whose column sums are dgamma
 * (dbeta = column sums of dy: mc_colsum_bf16); t_out may be NULL */
int mc_layernorm_bwd_bf16(const void* x, int64_t ldx, const void* g, const void* dy, int64_t ldy, void* dx, int64_t ldd, void* t_out,
                          int64_t ldt, int M, int D, float eps, void* stream);
int mc_swiglu_bwd_bf16(const void* gate_up, int64_t ld, const void* dinter, int64_t ldi, void* dgate_up, int64_t ldg, int M, int I,
                       void* stream);
int mc_act_bf16(const void* pre, const void* dy, void* out, int64_t n, int act, void* stream);   /* dy NULL: act(pre); else dy*act'(pre) */
int mc_ce_loss_f32(const float* logits, int64_t ld, const int64_t* labels, float* loss_rows, void* dlogits_bf16, int64_t ldd, int M,
                   int V, float inv_n, void* stream);
int mc_colsum_bf16(const void* x, int64_t ld, float* out, int M, int C, void* stream);
int mc_rope_inplace_bf16(void* x, int64_t ld, const int32_t* row_pos, const float* cos_table, const float* sin_table, int M,
                         int n_heads, int D, float sign, void* stream);
int mc_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16, int64_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* AdamW with two learning-rate groups chosen per element (llava_trainer.py:210-290: --mm_projector_lr / --mm_language_lr put the
 * lora_A.default / lora_B.default tensors into the projector group; run_finetune_vision_damc.sh:28).  The flat buffer is described by
 * chunks; element e of a chunk's tensor takes lr_alt when (idx0 + e) mod period < width (period 0: always lr).  off / n / idx0 / period /
 * width are multiples of 4 elements. */
typedef struct mc_adamw_seg { long long off; int n; int idx0; int period; int width; } mc_adamw_seg;
int mc_adamw_segments_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16, const mc_adamw_seg* segs_dev,
                          int n_segs, float lr, float lr_alt, float beta1, float beta2, float eps, float weight_decay, int step,
                          float grad_scale, void* stream);
/* y += alpha * x (fp32, flat): gradient accumulation over micro-batches (run_finetune_vision_damc.sh:45 --gradient_accumulation_steps) */
int mc_axpy_f32(float* y, const float* x, int64_t n, float alpha, void* stream);
int mc_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream);

/* ---- attention backward (training step; replaces autograd through multimodal_llama.py:295-312 and the flash-attn backward of
 * train/multimodal_flash_attn_monkey_patch.py:96-106).  MHA only (H == Hkv).  lse from mc_attn_prefill_lse_bf16; delta is
 * scratch [B, H, Lq] fp32.  Strides in elements, head_dim contiguous.                                                       */
typedef struct mc_attn_bwd_args {
    const void* q; int64_t q_sb, q_st, q_sh; const void* k; int64_t k_sb, k_st, k_sh; const void* v; int64_t v_sb, v_st, v_sh;
    const void* o; const void* d_o; int64_t o_sb, o_st, o_sh; const float* lse; float* delta;
    void* dq; int64_t dq_sb, dq_st, dq_sh; void* dk; int64_t dk_sb, dk_st, dk_sh; void* dv; int64_t dv_sb, dv_st, dv_sh;
    const int32_t* kv_lens; int B, H, Lq, S, D, causal, q_offset; float scale;
    float dropout_p; unsigned long long dropout_seed; unsigned int dropout_stream;   /* > 0: the forward was mc_attn_prefill_dropout_bf16 with
                                                                                      * these values (masks are regenerated, S % 4 == 0) */
} mc_attn_bwd_args;
int mc_attn_bwd_bf16(const mc_attn_bwd_args* args, void* stream);

/* ---- attention (multimodal_llama.py:295-312; CLIPAttention) ------------------------------------------- */
/* Optional last argument of the attention entry points (NULL: none of it).
 * key_valid [B][key_valid_stride] bytes, 0 = that key is masked for every query of the sequence - attention masks that are not "every key
 * below a length" (left padding, holes: the additive padding mask of LlamaModel._prepare_decoder_attention_mask,
 * multimodal_llama.py:543-545); kv_lens / causal keep their meaning (a key must pass all tests).
 * b_inner > 0: two-level batch index - batch entry b = (b / b_inner, b % b_inner) is addressed at (b / b_inner) * *_sb + (b % b_inner) *
 * inner_stride.  Only for prefill launches with Lq, S <= 8 and no relative-position table (the temporal attention of LanguageBind-Video,
 * modeling_video.py:105-130, over the (b t) n d layout: sequence (b, n), its t tokens a frame apart - attended in place instead of
 * through a permuted copy); any other launch is refused. */
typedef struct mc_attn_mask {
    const void* key_valid; int64_t key_valid_stride;
    int b_inner; int64_t inner_stride;
} mc_attn_mask;
int mc_attn_prefill_bf16(const void* q, int64_t q_sb, int64_t q_st, int64_t q_sh, const void* k, int64_t k_sb, int64_t k_st,
                         int64_t k_sh, const void* v, int64_t v_sb, int64_t v_st, int64_t v_sh, void* o,
                         int64_t o_row_stride, const int32_t* out_map, const int32_t* kv_lens, int B, int H, int Hkv, int Lq,
                         int S, int D, int causal, int q_offset, float scale, const float* rel_table, int rel_stride, int rel_off,
                         const float* q_gate, const mc_attn_mask* mask, void* stream);   /* rel_table/q_gate: gated relative-position bias of BEATs, may be null */
/* training forward with dropout on the attention probabilities (BertSelfAttention.dropout of the Q-Former projector,
 * multimodal_projector/Qformer.py:136, :259): O = (softmax(S) * keep / (1 - p)) V, lse of the un-dropped softmax; element
 * e = ((b H + h) Lq + q) S + key is kept iff word (e & 3) of Philox4x32-10(counter = (e >> 2, stream_id, 0), key = seed) >= p 2^32 */
int mc_attn_prefill_dropout_bf16(const void* q, int64_t q_sb, int64_t q_st, int64_t q_sh, const void* k, int64_t k_sb, int64_t k_st,
                                 int64_t k_sh, const void* v, int64_t v_sb, int64_t v_st, int64_t v_sh, void* o, int64_t o_row_stride,
                                 const int32_t* kv_lens, int B, int H, int Hkv, int Lq, int S, int D, int causal, int q_offset, float scale,
                                 float* lse, float dropout_p, unsigned long long seed, unsigned int stream_id, const mc_attn_mask* mask,
                                 void* stream);
/* as mc_attn_prefill_bf16, plus lse [B, H, Lq] fp32 = log2-sum-exp of the scaled scores (input of mc_attn_bwd_bf16) */
int mc_attn_prefill_lse_bf16(const void* q, int64_t q_sb, int64_t q_st, int64_t q_sh, const void* k, int64_t k_sb, int64_t k_st,
                             int64_t k_sh, const void* v, int64_t v_sb, int64_t v_st, int64_t v_sh, void* o,
                             int64_t o_row_stride, const int32_t* out_map, const int32_t* kv_lens, int B, int H, int Hkv, int Lq,
                             int S, int D, int causal, int q_offset, float scale, const float* rel_table, int rel_stride, int rel_off,
                             const float* q_gate, float* lse, const mc_attn_mask* mask, void* stream);
/* The attention probabilities themselves (output_attentions=True: `attn_weights` of LocalLoraAttention.forward, multimodal_llama.py:295-312 -
 * softmax in fp32 over the scaled, masked scores, cast to the model dtype): probs [B, H, Lq, S] bf16, row (b, h, i) = softmax_j(scale q_i.k_j)
 * over the keys j < kv_lens[b] (NULL: S) that pass the causal test j <= i + q_offset (causal != 0) and the optional key_valid [B][kv_stride]
 * bytes; masked entries are 0.  A debugging / analysis output, not on the generation path: one wave per (b, h, query) row, O(Lq S D). */
int mc_attn_probs_bf16(const void* q, int64_t q_sb, int64_t q_st, int64_t q_sh, const void* k, int64_t k_sb, int64_t k_st, int64_t k_sh,
                       const int32_t* kv_lens, const void* key_valid, int64_t kv_stride, void* probs, int B, int H, int Hkv, int Lq, int S,
                       int D, int causal, int q_offset, float scale, void* stream);
/* Decode attention (one query per sequence).  Batch-invariant by construction: the cached keys of a sequence are cut into chunks of 256 - a
 * function of the sequence's own length - every chunk's partial softmax is computed by one wave from a fresh state, and the partials are
 * combined in chunk order; which wave / workgroup takes which chunk changes no bit.  nsplit >= 1 = workgroups per head (group): a launch
 * shape for small batches, NOT part of the result.  workspace (mc_attn_decode_workspace_bytes for a cache of S positions) is needed when
 * nsplit > 1 or S > 19 968; NULL otherwise. */
int mc_attn_decode_workspace_bytes(int B, int H, int D, int S, int64_t* bytes);
int mc_attn_decode_bf16(const void* q, int64_t q_sb, int64_t q_sh, const void* k, int64_t k_sb, int64_t k_st, int64_t k_sh,
                        const void* v, int64_t v_sb, int64_t v_st, int64_t v_sh, void* o, int64_t o_sb, void* workspace,
                        const int32_t* kv_lens, int B, int H, int Hkv, int S, int D, int nsplit, float scale, const mc_attn_mask* mask,
                        void* stream);
/* decode attention with RoPE + KV-cache append fused in (multimodal_llama.py:281-312 for a one-token step): qkv [B, (H + 2 Hkv) * D]
 * is the pre-rotary q|k|v row of the token at position kv_lens[b] - 1; its rotated key / value are attended from registers and
 * written to the caches by the same launch.                                                                                */
int mc_attn_decode_rope_bf16(const void* qkv, int64_t qkv_ld, const float* cos_table, const float* sin_table, void* k_cache,
                             int64_t k_sb, int64_t k_st, int64_t k_sh, void* v_cache, int64_t v_sb, int64_t v_st, int64_t v_sh,
                             void* o, int64_t o_sb, void* workspace, const int32_t* kv_lens, int B, int H, int Hkv, int S, int D,
                             int nsplit, float scale, const mc_attn_mask* mask, void* stream);

/* ---- row kernels ---------------------------------------------------------------------------------- */
int mc_silu_mul_bf16(const void* gate_up, int64_t ld, void* out, int64_t ldo, int M, int I, void* stream);   /* :392-394 */
/* dst[b] = src[b * Lq + lens[b] - 1] (rows of D bf16): the last valid row of every sequence of a [B, Lq, D] tensor (the rotated query of
 * the last prompt token, for the last-layer tail of mc_llm_prefill) */
int mc_gather_last_rows_bf16(const void* src, int64_t ld_src, const int32_t* lens, int Lq, void* dst, int64_t ld_dst, int B, int D, void* stream);
int mc_copy_rows_bf16(const void* src, int64_t ld_src, const int32_t* src_idx, void* dst, int64_t ld_dst,
                      const int32_t* dst_idx, int n_rows, int D, void* stream);       /* splice: multimodal_arch.py:349-378 */
int mc_embed_rows_bf16(const void* table, int64_t ld_table, const int64_t* ids, void* dst, int64_t ld_dst,
                       const int32_t* dst_idx, int n_rows, int D, void* stream);      /* embed_tokens */
int mc_argmax_f32(const void* x, int64_t ld, int64_t* out, int M, int N, void* stream);  /* greedy_search argmax */
int mc_im2col_bf16(const void* in, void* out, int B, int C, int Hin, int Win, int kh, int kw, int sh, int sw, int Kp,
                   void* stream);                                                     /* patch-embed conv */
int mc_vit_assemble_bf16(const void* patches, const void* cls, const void* pos, void* out, int B, int T, int D, void* stream);
int mc_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);

/* ---- encoder-specific row kernels (csrc/encoders.hip) ------------------------------------------------ */
int mc_add_rows_bf16(const void* x, int64_t ldx, const void* table, int64_t ldt, const int32_t* idx, void* out, int64_t ldo,
                     int n_rows, int D, void* stream);       /* out[r] = x[r] + table[idx[r]]  (languagebind/video/modeling_video.py:110-113) */
int mc_zero_rows_bf16(void* x, int64_t ldx, const int32_t* rows, int n_rows, int D, void* stream);   /* beats/backbone.py:150-151 */
/* BEATs forward_padding_mask + x[padding_mask] = 0 on the device (beats/BEATs.py:120-132, beats/backbone.py:150-151): frame mask uint8
 * [B, mask_stride] (non-zero = padded frame; the first T * span frames count), token t of a clip is padded when all its `span` frames
 * are; kv_lens[b] = un-padded tokens, padded rows of x [B * T, ldx] are zeroed, bad_flag[0] |= 1 when a clip's padding is not a suffix. */
int mc_beats_padding_bf16(const void* frame_mask_u8, int64_t mask_stride, int B, int T, int span, void* x, int64_t ldx, int D,
                          int32_t* kv_lens, int32_t* bad_flag, void* stream);
int mc_im2col_ex_bf16(const void* in, int64_t s_b, int64_t s_c, int64_t s_h, int64_t s_w, void* out, int B, int C, int Hin, int Win,
                      int c0, int Cg, int kh, int kw, int sh, int sw, int ph, int pw, int oh, int ow, int Kp, void* stream);
int mc_beats_gate_f32(const float* g8, const float* grep_a, float* gate, int B, int L, int H, void* stream);   /* beats/backbone.py:689-697 */
int mc_group_max_bf16(const void* x, int64_t ldx, void* out, int64_t ldo, void* bcast, int64_t ldb, int G, int n, int C,
                      void* stream);                          /* pointbert/dvae.py:216-221 */
int mc_fps_bf16(const void* pts, int B, int N, int C, const int32_t* start_idx, int npoint, int32_t* out_idx, float* centers,
                void* stream);                                /* pointbert/misc.py:40-60 */
int mc_knn_group_bf16(const void* pts, int B, int N, int C, const float* centers, int G, int k, void* out, int Kp, int32_t* out_idx,
                      void* stream);                          /* pointbert/dvae.py:107-187 */
int mc_f32_rows_to_bf16(const float* in, int C, void* out, int Kp, int64_t rows, void* stream);

/* ---- device-resident greedy-loop state: [pos(B) | kvlen(B) | iota(B) | zeros(B) | step | pad(3)] int32 ---- */
int mc_decode_state_init(int32_t* state, const int32_t* prompt_lens, int B, int step0, void* stream);
int mc_decode_state_advance(int32_t* state, int B, void* stream);
int mc_argmax_step_f32(const void* x, int64_t ld, int64_t* next_ids, int64_t* out_ids, int64_t ld_out, const int32_t* step_ptr,
                       int M, int N, void* stream);
/* Sampled step (generate(do_sample=True, temperature, top_p[, top_k]): eval/model_multimodal_qa_loader.py:94-102, serve/model_worker.py:160-185;
 * warper semantics of transformers 4.31 generation/logits_process.py): scores = logits / temperature; keep scores >= the top_k-th largest
 * (top_k = 0: off); drop the ascending prefix whose cumulative softmax mass <= 1 - top_p (top_p = 1: off; the largest always stays);
 * draw from the renormalised rest by inverse CDF with one Philox4x32-10 uniform per (seed, row, step).  step / seed come from device memory
 * (step_ptr, seed_ptr[2]) when given, else from step_const / seed_const.  uniform_in [M] (optional) replaces the RNG; probs_out [M][N]
 * (optional, row stride ldp) receives the final probabilities.  next_ids[m] = token; out_ids[m*ld_out + step] = token when out_ids != NULL. */
int mc_sample_step_f32(const float* logits, int64_t ld, int64_t* next_ids, int64_t* out_ids, int64_t ld_out, const int32_t* step_ptr,
                       int step_const, const uint32_t* seed_ptr, unsigned long long seed_const, int M, int N, float temperature, int top_k,
                       float top_p, const float* uniform_in, float* probs_out, int64_t ldp, void* stream);

/* out[m][n] = logits[m][n] - logsumexp_n(logits[m]) (fp32): the scoring step of beam search (generate(num_beams > 1): transformers 4.31
 * beam_search, forwarded by eval/model_multimodal_qa_loader.py:94-102) */
int mc_log_softmax_f32(const float* logits, int64_t ld, float* out, int64_t ldo, int M, int N, void* stream);

/* ---- checkpoint files (csrc/ckpt_reader.cpp) --------------------------------------------------------------------------------
 * Replaces torch.load / safetensors in load_pretrained_model (model/builder.py:148, :157-168): adapter_model.bin, non_lora_trainables.bin,
 * mm_projector.bin, the base model's (sharded) pytorch_model-*.bin or *.safetensors, encoder checkpoints.  mc_ckpt_open maps the file and
 * indexes its tensors: torch zip archives (STORED zip / zip64 + a protocol-2 pickle read by a restricted interpreter that never calls
 * anything) and safetensors.  Every tensor is (name, dtype code, shape, strides in elements, pointer into the mapping); nested containers
 * are flattened with '.'-joined names.  Pointers stay valid until mc_ckpt_close.  Legacy non-zip torch files are refused (error 1);
 * every size, offset and count read from the file is bounds-checked with overflow-safe arithmetic, object trees deeper than 64 levels
 * (or self-referential through the pickle memo) are refused.                                                                          */
#define MC_CKPT_F32 0
#define MC_CKPT_F16 1
#define MC_CKPT_BF16 2
#define MC_CKPT_F64 3
#define MC_CKPT_I64 4
#define MC_CKPT_I32 5
#define MC_CKPT_I16 6
#define MC_CKPT_I8 7
#define MC_CKPT_U8 8
#define MC_CKPT_BOOL 9
int mc_ckpt_open(const char* path, void** handle);
int mc_ckpt_close(void* handle);
int mc_ckpt_count(void* handle, int* n_tensors);
/* storage_bytes = bytes from `data` to the end of the tensor's storage record (strided views may touch all of them) */
int mc_ckpt_entry(void* handle, int index, const char** name, int* dtype, int* ndim, const int64_t** shape, const int64_t** strides,
                  const void** data, int64_t* storage_bytes);
/* contiguous tensors only: hipMemcpyAsync from the mapped file to dst_device (numel * element size bytes) on `stream` */
int mc_ckpt_copy_to_device(void* handle, int index, void* dst_device, void* stream);
/* Non-tensor leaves of the object tree (None / bool / int / float / str): encoder checkpoints keep their configuration next to the
 * weights - BEATs files are {'cfg': {...}, 'model': state_dict} (beats/BEATs.py:120-148 reads checkpoint['cfg']).  `name` is the
 * '.'-joined key as for tensors; `path` (also mc_ckpt_entry_path for tensors) joins the levels with 0x1f and marks list / tuple
 * indices with a leading 0x1e, so a caller can rebuild the exact tree even when keys contain dots.  svalue is NOT NUL-terminated
 * text in general: use slen.  safetensors files have no scalars.                                                                     */
#define MC_CKPT_NONE 0
#define MC_CKPT_BOOLEAN 1
#define MC_CKPT_INT 2
#define MC_CKPT_FLOAT 3
#define MC_CKPT_STR 4
#define MC_CKPT_EMPTY_LIST 5   /* an empty list / tuple / dict keeps its place in the tree (BEATs cfg entries such as [] or ()) */
#define MC_CKPT_EMPTY_TUPLE 6
#define MC_CKPT_EMPTY_DICT 7
int mc_ckpt_entry_path(void* handle, int index, const char** path);
int mc_ckpt_scalar_count(void* handle, int* n_scalars);
int mc_ckpt_scalar(void* handle, int index, const char** name, const char** path, int* kind, int64_t* ivalue, double* fvalue,
                   const char** svalue, int64_t* slen);

/* ---- composed Vicuna backbone runtime (csrc/llm_runtime.cpp) -------------------------------------------
 * Replaces MultimodalLlamaModel.forward + lm_head (model/language_model/multimodal_llama.py:488-619, :720) and the
 * greedy loop driven by model.generate (eval/model_multimodal_qa_loader.py:94-102).  The handle owns only host
 * tables of borrowed device pointers; KV cache, workspace and decode state are caller-owned device buffers.      */
typedef struct mc_llm_config {
    int hidden, inter, n_layers, n_heads, n_kv_heads, head_dim, vocab, n_adapters, max_pos;
    float rms_eps;
} mc_llm_config;

int mc_llm_create(const mc_llm_config* cfg, void** handle);
int mc_llm_destroy(void* handle);
/* layer_w[(layer*n_adapters + adapter)*4 + {0:qkv,1:o,2:gate_up,3:down}] = packed weights composed by
 * mc_compose_weight_ex_bf16: q|k|v rows fused with input_layernorm's weight folded into the columns; gate|up rows interleaved
 * per 16-row block (gate even, up odd) with post_attention_layernorm's weight folded in; o_proj / down_proj plain.
 * final_norm: bf16 [hidden] (model.norm, applied unfolded before lm_head); cos/sin tables [max_pos, head_dim/2] fp32
 * (LlamaRotaryEmbedding of transformers 4.31, computed in fp32).                                                        */
int mc_llm_set_weights(void* handle, const void* const* layer_w, const void* final_norm, const void* lm_head_packed,
                       const void* embed_table, const float* cos_table, const float* sin_table);
/* options: "use_graph" (decode steps replayed from a hipGraph, default 1); "graph_logits" (default 0; 1: a mc_llm_decode call with logits_out
 * still replays the graph and copies every step's logits out of the workspace between replays - the logits all-gather of an eval loop);
 * "profile" (below); "tail_adapter" (default -1 = off; a >= 0: the
 * NEXT mc_llm_prefill call, if it asks for last-row logits / next ids only - hidden_out null - runs the LAST layer's attention, o_proj and
 * MLP for the last token of every sequence only, with adapter a's weights; the layer's q|k|v projection still covers every row, so the
 * KV cache is what the all-rows path writes.  ONE-SHOT: the value is the caller's promise that every last_rows entry of that batch lies
 * in a group routed to adapter a (last_rows is device memory, the library cannot check it without a synchronising copy); the call consumes
 * it and resets the option to -1, so the promise never carries over to another batch) */
int mc_llm_set_option(void* handle, const char* name, int value);
/* read back: "use_graph"; "graph_active" = 1 when the last mc_llm_decode replayed a hipGraph (0: one launch per kernel); "graph_captures" /
 * "graph_failures" = decode-step graphs captured / capture attempts that failed since mc_llm_create */
int mc_llm_get_option(void* handle, const char* name, int* value);
/* "profile" option (mc_llm_set_option): while 1, every launch of the layer loop is bracketed by HIP events on its stream (profiled decode
 * calls run one launch per kernel, no graph).  mc_llm_profile_read sums them per kernel class - index 0 q|k|v GEMM, 1 RoPE + cache append
 * (prefill), 2 attention, 3 o_proj GEMM, 4 RMS factor (prefill), 5 gate|up GEMM + SwiGLU, 6 down_proj GEMM, 7 lm_head GEMM, 8 other - for
 * phase 0 (prefill) or 1 (decode); arrays of mc_llm_profile_kinds() entries.  Enabling the option clears the records.                  */
int mc_llm_profile_kinds(void);
int mc_llm_profile_read(void* handle, int phase, double* total_ms, int64_t* launches);
/* next-token rule of mc_llm_decode: do_sample = 0 greedy arg-max (default), 1 = mc_sample_step_f32 with these parameters and the seed the
 * caller stored at state[4B+1], state[4B+2] */
int mc_llm_set_sampling(void* handle, int do_sample, float temperature, int top_k, float top_p);
/* One-shot key mask of the NEXT mc_llm_prefill or mc_llm_decode call (handed to every layer's attention of that call as its mc_attn_mask.key_valid): the
 * batch's attention mask when it is not a suffix mask - left-padded batches, masks with holes.  key_valid [B][row_stride >= Smax] bytes over
 * CACHE positions (generated positions must be 1).  A decode call with a mask runs one launch per kernel (no graph replay). */
int mc_llm_set_key_mask(void* handle, const void* key_valid, int64_t row_stride);
/* One-shot capture for the NEXT mc_llm_prefill call (forward(output_hidden_states / output_attentions), multimodal_llama.py:561-604, :676-688):
 * hidden_snapshots (may be NULL) receives n_layers + 1 copies of the routed hidden state [M, hidden] bf16 - index 0 the input embeddings,
 * index l + 1 the output of decoder layer l (un-normed; the reference's last tuple entry is the final norm of index n_layers, which the
 * call returns in hidden_out); q_snapshots (may be NULL) receives every layer's rotated queries [B, Lq, H, D] bf16 in sequence order (with
 * the layer's keys, which stay in the KV cache, they give the attention probabilities: mc_attn_probs_bf16).  A capturing call runs every
 * layer for every row (no last-layer tail). */
int mc_llm_set_capture(void* handle, void* hidden_snapshots, void* q_snapshots);
int mc_llm_workspace_bytes(void* handle, int M, int B, int Lq, int64_t* bytes);
/* Prefill over M rows in routed order.  group_start / group_adapter are HOST arrays (n_groups+1 / n_groups); the other
 * int32 arrays are device arrays: row_b/row_pos/row_t [M], out_map [B*Lq] (sequence slot -> routed row, -1 = padding),
 * kv_lens [B], last_rows [B] (routed row of each sample's last token).  x_routed is updated in place (with "tail_adapter" set and
 * hidden_out null its rows hold the input of the last layer, not its output: only the last tokens' outputs exist, in the workspace). */
int mc_llm_prefill(void* handle, void* x_routed, int M, int n_groups, const int32_t* group_start, const int32_t* group_adapter,
                   const int32_t* row_b, const int32_t* row_pos, const int32_t* row_t, const int32_t* out_map,
                   const int32_t* kv_lens, const int32_t* last_rows, int B, int Lq, void* k_cache, void* v_cache, int Smax,
                   void* workspace, void* hidden_out, float* logits_out, int64_t* next_ids, void* stream);
/* Greedy / sampled decode of n_steps tokens on the device.  kv_len_max (host value) = the largest number of keys any sequence already has in
 * the cache (prompt length + tokens decoded so far); kv_len_max + n_steps > Smax is refused (error 1).  n_groups of mc_llm_prefill <= 64.
 * With "use_graph" the step is captured once per buffer set and replayed from a hipGraph; stream 0 (the legacy null stream, on which HIP
 * refuses capture) is served by a handle-owned stream ordered against stream 0 with events - see mc_llm_get_option("graph_active").       */
int mc_llm_decode(void* handle, int B, int n_steps, int64_t* next_ids, int64_t* out_ids, int64_t ld_out, int32_t* state,
                  void* k_cache, void* v_cache, int Smax, int kv_len_max, void* workspace, float* logits_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
