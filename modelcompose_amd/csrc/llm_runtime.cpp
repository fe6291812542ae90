// Host-side runtime of the composed Vicuna backbone: the 32-layer loop, the routed (per-adapter) GEMM
// groups and the device-resident greedy loop, with no Python between kernels.
//
// Mirrors, on the device, modelcompose/model/language_model/multimodal_llama.py:
//   MultimodalLlamaModel.forward (:488-619)  -> mc_llm_prefill / mc_llm_decode layer loops
//   MultimodalLlamaDecoderLayer.forward (:408-468), LocalLoraAttention (:210-342), LocalLoraMLP (:363-396)
//   lm_head (:720) + greedy argmax of transformers' greedy_search (model_multimodal_qa_loader.py:94-102)
// LocalLoRA routing: hidden rows are kept grouped by adapter ("routed order"); every token has exactly one
// active adapter (multimodal_arch.py:452-453), so each group runs ONE dense GEMM against that adapter's
// pre-composed weight (mc_compose_weight_ex_bf16) instead of all adapters on all tokens + mask-sum (:262-268).
// During decode the modal mask is dropped (:435-438): adapter 0 ('default') only.
// Fusions (results-identical restructuring of MultimodalLlamaDecoderLayer.forward :408-468):
//   * RMSNorm = (per-row 1/rms) x (per-column weight): the weight is folded into the q|k|v and gate|up columns at compose
//     time, the 1/rms factor is a GEMM epilogue input (row_scale) -> no normalised copy of the hidden state is written;
//   * silu(gate) * up is the epilogue of the gate|up GEMM (weights interleaved per 16 rows) -> no [M, 2I] intermediate;
//   * decode steps and the last-token tail of a prefill run the strip GEMM family (gemm_strip.hip) whatever the batch size: the RMS factor is
//     computed from the x fragments the GEMM streams, o_proj / down_proj add the residual in their epilogue - 5 launches per layer, no
//     normalisation pass, no split-K slabs.  Together with the per-sequence chunk schedule of the decode attention a sequence's logits and
//     tokens are bit-identical whatever batch it is decoded in.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../include/mc_hip.h"

void mc_set_error(const char* fmt, ...);

namespace {

struct Llm {
    mc_llm_config cfg;
    std::vector<const void*> layer_w;       // [(layer*n_adapters + adapter)*4 + {qkv,o,gate_up,down}]
    const void* final_norm = nullptr;
    const void* lm_head = nullptr;
    const void* embed = nullptr;
    const float* cos_t = nullptr;
    const float* sin_t = nullptr;
    bool weights_set = false;
    // decode graph cache
    // decode-step graphs, keyed by every pointer / parameter baked into the captured launches; a few entries so that two generation
    // pipelines (their own KV cache, workspace and state each) can alternate without re-capturing
    struct Key { int B, Smax; const void *kc, *vc, *ws, *state, *next_ids, *out_ids, *logits; int64_t ld_out; int sample, top_k; float temp, top_p; };
    static constexpr int kGraphs = 4;
    hipGraphExec_t graph_exec[kGraphs] = {nullptr, nullptr, nullptr, nullptr};
    Key gkey[kGraphs] = {};
    int graph_next = 0;                      // round-robin victim
    bool use_graph = true;
    const void* key_mask = nullptr;          // one-shot (mc_llm_set_key_mask): the next prefill / decode call's attention key mask
    int64_t key_mask_stride = 0;
    const void* call_key_mask = nullptr;     // ... while that call runs
    int64_t call_key_mask_stride = 0;
    bool graph_logits = false;               // "graph_logits" option: a decode call that asks for step logits still replays the graph (the step's
                                             // logits are copied out of the workspace between replays) instead of one launch per kernel
    // one-shot capture of the NEXT prefill (mc_llm_set_capture): cap_hidden receives n_layers + 1 snapshots of the routed hidden state
    // (the embeddings, then every layer's output), cap_q every layer's rotated queries [B, Lq, H, D] - what forward(output_hidden_states /
    // output_attentions) of the reference returns is built from them (multimodal_llama.py:561-604)
    void* cap_hidden = nullptr;
    void* cap_q = nullptr;
    int tail_adapter = -1;                   // >= 0: generate()'s prefill runs the last layer's attention + MLP for the last token of every
                                             // sequence only (all of them routed to this adapter); -1: every row (forward(), mixed adapters)
    // hipStreamBeginCapture is refused on the legacy null stream (torch's default current stream).  Callers that pass stream 0 have their
    // decode graphs captured and replayed on this handle-owned non-blocking stream, ordered against stream 0 by the two events.
    hipStream_t own_stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // observability (mc_llm_get_option): did the last mc_llm_decode replay a graph; captures made; captures that failed
    int graph_active = 0, graph_captures = 0, graph_failures = 0;
    bool capture_warned = false;
    // live per-kernel-class timing ("profile" option; bench.py's roofline_decode object): HIP events recorded on the launch stream around
    // every launch of the layer loop while enabled.  Profiled decode calls run one launch per kernel (events are not captured into graphs).
    struct PRec { int phase, kind; hipEvent_t a, b; };
    bool prof_on = false;
    std::vector<PRec> prof;
    // next-token rule: greedy arg-max, or temperature / top-k / top-p sampling (seed: 2 x u32 at state[4B+1] on the device)
    bool do_sample = false;
    float temperature = 1.0f, top_p = 1.0f;
    int top_k = 0;
};

constexpr int kMaxGroups = 64;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Ws {          // carve-up of the caller-provided workspace
    char *qkv, *qseq, *attn, *inter, *xl, *nl, *logits;
    float* rs;
    int64_t ldx, lda, ldi;     // row strides (elements) of xl / nl, attn, inter
    size_t total;
};

// Row strides of the decode buffers: plain widths.  (Rows 128 bytes longer than their width - to spread the 16 rows of an MFMA fragment
// over L2 channels - measured no different for the strip GEMMs: 14.2 / 32.1 / 23.4 / 36.3 us against 14.5 / 31.9 / 23.5 / 36.5 us for
// o / down / q|k|v / gate|up at 48 rows, profiles/r06_probes/strip_check.json.)
constexpr int kRowPad = 0;

Ws carve(const mc_llm_config& c, int M, int B, int Lq, char* base, bool decode) {
    const size_t hd = c.hidden, qkvd = (size_t)(c.n_heads + 2 * c.n_kv_heads) * c.head_dim;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align_up(bytes, 256); return p; };
    Ws w;
    w.ldx = (int64_t)hd + kRowPad;
    w.lda = (int64_t)hd + (decode ? kRowPad : 0);
    w.ldi = (int64_t)c.inter + (decode ? kRowPad : 0);
    w.qkv = take((size_t)M * qkvd * 2);
    w.qseq = take((size_t)B * Lq * c.n_heads * c.head_dim * 2);
    w.attn = take((size_t)M * w.lda * 2);
    w.inter = take((size_t)M * w.ldi * 2);
    w.rs = (float*)take((size_t)M * 4);
    w.xl = take((size_t)B * w.ldx * 2);
    w.nl = take((size_t)B * w.ldx * 2);
    w.logits = take((size_t)B * c.vocab * 4);
    w.total = off;
    return w;
}

inline int c_max_smax(const mc_llm_config& c) { return c.max_pos + 64; }
// partials of the decode attention for a cache of up to max_pos (+ one 64-position rounding) keys
int64_t attn_ws_bytes(const mc_llm_config& c, int B) {
    int64_t b = 0;
    mc_attn_decode_workspace_bytes(B, c.n_heads, c.head_dim, c_max_smax(c), &b);
    return b;
}

#define RUN(call)                    \
    do {                             \
        int rc__ = (call);           \
        if (rc__ != 0) return rc__;  \
    } while (0)

// kernel classes of the layer loop (mc_llm_profile_read)
enum { PK_QKV = 0, PK_ROPE, PK_ATTN, PK_O, PK_RMS, PK_GATE_UP, PK_DOWN, PK_HEAD, PK_OTHER, PK_COUNT };

// RUN with optional event bracketing on the launch stream
#define RUNP(m_, phase_, kind_, stream_, call)                                         \
    do {                                                                               \
        hipEvent_t pa__ = nullptr, pb__ = nullptr;                                     \
        if ((m_)->prof_on) {                                                           \
            (void)hipEventCreate(&pa__); (void)hipEventCreate(&pb__);                  \
            (void)hipEventRecord(pa__, (hipStream_t)(stream_));                        \
        }                                                                              \
        int rc__ = (call);                                                             \
        if ((m_)->prof_on) {                                                           \
            (void)hipEventRecord(pb__, (hipStream_t)(stream_));                        \
            (m_)->prof.push_back({(phase_), (kind_), pa__, pb__});                     \
        }                                                                              \
        if (rc__ != 0) return rc__;                                                    \
    } while (0)

int check_handle(Llm* m, const char* fn) {
    if (!m) { mc_set_error("%s: null handle", fn); return 1; }
    if (!m->weights_set) { mc_set_error("%s: mc_llm_set_weights has not been called", fn); return 1; }
    return 0;
}

int gemm_ex(const void* x, int64_t ldx, const void* w, const void* residual, int64_t ldr, void* out, int64_t ldo, int M, int N, int K,
            int out_f32, const float* row_scale, int swiglu, int split_k, void* stream) {
    mc_gemm_args a;
    a.x = x; a.ldx = ldx; a.w_packed = w; a.bias = nullptr; a.residual = residual; a.ldr = ldr; a.out = out; a.ldo = ldo;
    a.M = M; a.N = N; a.K = K; a.act = MC_ACT_NONE; a.out_f32 = out_f32; a.alpha = 1.0f; a.beta = 1.0f;
    a.row_scale = row_scale; a.swiglu = swiglu; a.split_k = split_k; a.rms_eps = 0.f; a.rope = nullptr; a.rms_out = nullptr; a.rms_out_eps = 0.f;
    a.family = MC_GEMM_AUTO;
    return mc_gemm_ex_bf16(&a, stream);
}

int gemm_grouped(const void* x, int64_t ldx, const void* residual, int64_t ldr, void* out, int64_t ldo, int N, int K, const float* row_scale,
                 int swiglu, int n_groups, const int32_t* gstart, const void* const* weights, void* stream, float rms_eps = 0.f,
                 const mc_rope_scatter* rope = nullptr, float* rms_out = nullptr, float rms_out_eps = 0.f, int family = MC_GEMM_TILE) {
    mc_gemm_args a;
    a.x = x; a.ldx = ldx; a.w_packed = nullptr; a.bias = nullptr; a.residual = residual; a.ldr = ldr; a.out = out; a.ldo = ldo;
    a.M = 0; a.N = N; a.K = K; a.act = MC_ACT_NONE; a.out_f32 = 0; a.alpha = 1.0f; a.beta = 1.0f;
    a.row_scale = row_scale; a.swiglu = swiglu; a.split_k = 1; a.rms_eps = rms_eps; a.rope = rope; a.rms_out = rms_out; a.rms_out_eps = rms_out_eps;
    a.family = family;
    return mc_gemm_grouped_bf16(&a, n_groups, gstart, weights, stream);
}

// one decoder layer over rows grouped by adapter; x is updated in place.  On entry w.rs holds 1/rms of every row of x
// (input_layernorm's factor); on exit it holds the factor for the next layer's input_layernorm (or the final norm).
int layer_forward(Llm* m, int layer, char* x, int64_t ldh, int M, int n_groups, const int32_t* gstart, const int32_t* gadapter,
                  const Ws& w, const int32_t* row_b, const int32_t* row_pos, const int32_t* row_t, const int32_t* out_map,
                  const int32_t* kv_lens, int B, int Lq, char* kc, char* vc, int Smax, bool decode, int nsplit, void* attn_ws,
                  void* stream, int stage = 0, int phase_tag = -1) {
    // stage 0: the whole layer; 1: the q|k|v projection (+ RoPE / cache scatter) only; 2: everything after the attention (w.attn holds its
    // output) - the two halves of the last prefill layer of generate(), which runs the second half for the last tokens only
    const mc_llm_config& c = m->cfg;
    const int64_t hd = c.hidden, D = c.head_dim, H = c.n_heads, Hkv = c.n_kv_heads, I = c.inter;
    const int64_t qkvd = (H + 2 * Hkv) * D;
    const int64_t lda = w.lda, ldi = w.ldi;
    const size_t kv_layer = (size_t)B * Hkv * Smax * D * 2;
    char* kcl = kc + (size_t)layer * kv_layer;
    char* vcl = vc + (size_t)layer * kv_layer;
    const float scale = 1.0f / sqrtf((float)D);
    // Kernel family (mc_gemm_args.family): decode steps and the last-token tail of a prefill run the STRIP family whatever the batch size
    // (slices of 64 rows above 64), everything else the TILE family whatever the row count - a sequence's rows go through the same fp32
    // summation order in a batch of 1 and in a batch of 48 (bitwise batch invariance; model_multimodal_qa_loader.py:25-46 batches a
    // question differently at 1 and at 8 GPUs).
    // Strip launches: no normalisation pass at all - the GEMMs that read the hidden state compute its 1/rms from the x fragments they
    // stream (rms_eps), o_proj / down_proj add the residual in their epilogue: 5 launches per layer.
    const bool strip = decode;                                      // decode steps and the last-token tail: at most 64 rows per launch
    const int fam = strip ? MC_GEMM_STRIP : MC_GEMM_TILE;
    const float* rs_in = strip ? nullptr : w.rs;
    const float eps_in = strip ? c.rms_eps : 0.f;
    auto W = [&](int adapter, int which) { return m->layer_w[((size_t)layer * c.n_adapters + adapter) * 4 + which]; };

    // weights of every group for linear `which` (n_groups <= kMaxGroups is checked by mc_llm_prefill; decode has one group)
    const void* wg[kMaxGroups];
    auto W_all = [&](int which) {
        for (int g = 0; g < n_groups; ++g) wg[g] = W(gadapter[g], which);
        return (const void* const*)wg;
    };
    const int ph = phase_tag >= 0 ? phase_tag : (decode ? 1 : 0);      // profile phase: the prefill's last-token tail uses decode kernels but is prefill time
    // q|k|v = (x / rms) . (W_qkv diag(g_in))^T                                                  (:440-443, :262-268)
    // prefill: RoPE, the q re-ordering and the cache append are the projection's epilogue (mc_rope_scatter; a separate mc_rope_kv_bf16
    // launch inside the library when the launch is too small for the 256x256 kernel or the head size is not 128)            (:281-312)
    mc_rope_scatter rope{row_b, row_pos, row_t, m->cos_t, m->sin_t, w.qseq, kcl, vcl, (int)H, (int)Hkv, (int)D, Lq, Smax};
    if (stage != 2)
        RUNP(m, ph, PK_QKV, stream, gemm_grouped(x, ldh, nullptr, 0, w.qkv, qkvd, (int)qkvd, (int)hd, rs_in, 0, n_groups, gstart, W_all(0), stream, eps_in,
                                                 decode ? nullptr : &rope, nullptr, 0.f, fam));
    if (stage == 1) return 0;
    const mc_attn_mask kmask{m->call_key_mask, m->call_key_mask_stride, 0, 0};
    const mc_attn_mask* mk = m->call_key_mask ? &kmask : nullptr;
    if (stage == 2) {
    } else if (decode) {
        // one token per sequence (row b = sequence b): RoPE, the cache append and the attention are one launch (:281-312)
        RUNP(m, ph, PK_ATTN, stream, mc_attn_decode_rope_bf16(w.qkv, qkvd, m->cos_t, m->sin_t, kcl, Hkv * Smax * D, D, (int64_t)Smax * D, vcl,
                                     Hkv * Smax * D, D, (int64_t)Smax * D, w.attn, lda, attn_ws, kv_lens, B, (int)H, (int)Hkv, Smax, (int)D, nsplit,
                                     scale, mk, stream));
    } else {
        RUNP(m, ph, PK_ATTN, stream, mc_attn_prefill_bf16(w.qseq, (int64_t)Lq * H * D, H * D, D, kcl, Hkv * Smax * D, D, (int64_t)Smax * D, vcl,
                                     Hkv * Smax * D, D, (int64_t)Smax * D, w.attn, lda, out_map, kv_lens, B, (int)H, (int)Hkv, Lq, Smax,
                                     (int)D, 1, 0, scale, nullptr, 0, 0, nullptr, mk, stream));
    }
    // x += o_proj(attn)  (:447);  then 1/rms of the new x for post_attention_layernorm (:462)
    // (prefill: the factor of post_attention_layernorm comes out of the same launch - rms_out - instead of a pass over the new x)
    RUNP(m, ph, PK_O, stream, gemm_grouped(w.attn, lda, x, ldh, x, ldh, (int)hd, (int)hd, nullptr, 0, n_groups, gstart, W_all(1), stream, 0.f, nullptr,
                                           strip ? nullptr : w.rs, c.rms_eps, fam));
    // inter = silu(gate) * up with gate|up = (x / rms) . (W_gu diag(g_post))^T                 (:380-390)
    RUNP(m, ph, PK_GATE_UP, stream, gemm_grouped(x, ldh, nullptr, 0, w.inter, ldi, (int)(2 * I), (int)hd, rs_in, 1, n_groups, gstart, W_all(2), stream, eps_in,
                                                 nullptr, nullptr, 0.f, fam));
    // x += down_proj(inter)  (:466);  then 1/rms for the next layer's input_layernorm / the final norm
    RUNP(m, ph, PK_DOWN, stream, gemm_grouped(w.inter, ldi, x, ldh, x, ldh, (int)hd, (int)I, nullptr, 0, n_groups, gstart, W_all(3), stream, 0.f, nullptr,
                                              strip ? nullptr : w.rs, c.rms_eps, fam));
    return 0;
}

// final norm + lm_head over the B rows of xl (stride w.ldx); strip family whatever B is (one summation order per row, see layer_forward)
int head_forward(Llm* m, const char* xl, int B, const Ws& w, float* logits, void* stream, int phase) {
    const mc_llm_config& c = m->cfg;
    RUNP(m, phase, PK_OTHER, stream, mc_rmsnorm_bf16(xl, w.ldx, m->final_norm, w.nl, w.ldx, B, c.hidden, c.rms_eps, stream));
    mc_gemm_args a;
    a.x = w.nl; a.ldx = w.ldx; a.w_packed = m->lm_head; a.bias = nullptr; a.residual = nullptr; a.ldr = 0; a.out = logits; a.ldo = c.vocab;
    a.M = B; a.N = c.vocab; a.K = c.hidden; a.act = MC_ACT_NONE; a.out_f32 = 1; a.alpha = 1.0f; a.beta = 1.0f;
    a.row_scale = nullptr; a.swiglu = 0; a.split_k = 1; a.rms_eps = 0.f; a.rope = nullptr; a.rms_out = nullptr; a.rms_out_eps = 0.f;
    a.family = MC_GEMM_STRIP;
    RUNP(m, phase, PK_HEAD, stream, mc_gemm_ex_bf16(&a, stream));
    return 0;
}

// Workgroups per (b, h) of the decode attention: enough waves to fill the chip at small batches.  Results do not depend on it (the
// kernel's chunk schedule is a function of each sequence's own length): this is a launch-shape choice only.
int decode_nsplit(const mc_llm_config& c, int B, int Smax) {
    const int64_t waves = (int64_t)B * c.n_heads * 4;
    if (waves >= 2048) return 1;
    const int chunks = (Smax + 511) / 512;
    const int64_t want = (2048 + waves - 1) / waves;
    const int ns = (int)(want < chunks ? want : chunks);
    return ns > 0 ? ns : 1;
}

}  // namespace

extern "C" int mc_llm_create(const mc_llm_config* cfg, void** handle) {
    if (!cfg || !handle) { mc_set_error("mc_llm_create: null argument"); return 1; }
    if (cfg->hidden <= 0 || cfg->hidden % 64 || cfg->inter % 64 || cfg->n_layers <= 0 || cfg->n_adapters <= 0 ||
        cfg->n_heads % cfg->n_kv_heads || (cfg->head_dim != 64 && cfg->head_dim != 128) ||
        cfg->n_heads * cfg->head_dim != cfg->hidden || cfg->vocab % 4) {
        mc_set_error("mc_llm_create: unsupported geometry hidden=%d inter=%d heads=%d kv=%d head_dim=%d vocab=%d", cfg->hidden,
                     cfg->inter, cfg->n_heads, cfg->n_kv_heads, cfg->head_dim, cfg->vocab);
        return 1;
    }
    Llm* m = new Llm();
    m->cfg = *cfg;
    *handle = m;
    // the tile GEMMs' rms_out route uses a per-stream scratch pool: make sure it exists before the first graph capture (best effort)
    (void)mc_gemm_reserve_workspace(nullptr);
    return 0;
}

extern "C" int mc_llm_destroy(void* handle) {
    Llm* m = (Llm*)handle;
    if (!m) return 0;
    for (int i = 0; i < Llm::kGraphs; ++i)
        if (m->graph_exec[i]) (void)hipGraphExecDestroy(m->graph_exec[i]);
    if (m->own_stream) { (void)hipStreamSynchronize(m->own_stream); (void)mc_gemm_release_workspace(m->own_stream); (void)hipStreamDestroy(m->own_stream); }
    if (m->ev_in) (void)hipEventDestroy(m->ev_in);
    if (m->ev_out) (void)hipEventDestroy(m->ev_out);
    for (auto& r : m->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    delete m;
    return 0;
}

extern "C" int mc_llm_set_weights(void* handle, const void* const* layer_w, const void* final_norm, const void* lm_head_packed,
                                  const void* embed_table, const float* cos_table, const float* sin_table) {
    Llm* m = (Llm*)handle;
    if (!m || !layer_w || !final_norm || !lm_head_packed || !embed_table || !cos_table || !sin_table) {
        mc_set_error("mc_llm_set_weights: null argument");
        return 1;
    }
    const size_t n = (size_t)m->cfg.n_layers * m->cfg.n_adapters * 4;
    m->layer_w.assign(layer_w, layer_w + n);
    for (size_t i = 0; i < n; ++i)
        if (!m->layer_w[i]) { mc_set_error("mc_llm_set_weights: null weight pointer at index %zu", i); return 1; }
    m->final_norm = final_norm; m->lm_head = lm_head_packed; m->embed = embed_table;
    m->cos_t = cos_table; m->sin_t = sin_table;
    m->weights_set = true;
    for (int i = 0; i < Llm::kGraphs; ++i)
        if (m->graph_exec[i]) { (void)hipGraphExecDestroy(m->graph_exec[i]); m->graph_exec[i] = nullptr; }
    return 0;
}

extern "C" int mc_llm_set_option(void* handle, const char* name, int value) {
    Llm* m = (Llm*)handle;
    if (!m || !name) { mc_set_error("mc_llm_set_option: null argument"); return 1; }
    if (!strcmp(name, "use_graph")) { m->use_graph = value != 0; return 0; }
    if (!strcmp(name, "graph_logits")) { m->graph_logits = value != 0; return 0; }
    if (!strcmp(name, "tail_adapter")) {
        if (value >= m->cfg.n_adapters) { mc_set_error("mc_llm_set_option: tail_adapter %d of %d adapters", value, m->cfg.n_adapters); return 1; }
        m->tail_adapter = value < 0 ? -1 : value;
        return 0;
    }
    if (!strcmp(name, "profile")) {
        if (value && !m->prof_on) {
            for (auto& r : m->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
            m->prof.clear();
        }
        m->prof_on = value != 0;
        return 0;
    }
    mc_set_error("mc_llm_set_option: unknown option '%s'", name);
    return 1;
}

extern "C" int mc_llm_get_option(void* handle, const char* name, int* value) {
    Llm* m = (Llm*)handle;
    if (!m || !name || !value) { mc_set_error("mc_llm_get_option: null argument"); return 1; }
    if (!strcmp(name, "use_graph")) { *value = m->use_graph ? 1 : 0; return 0; }
    if (!strcmp(name, "graph_active")) { *value = m->graph_active; return 0; }
    if (!strcmp(name, "graph_captures")) { *value = m->graph_captures; return 0; }
    if (!strcmp(name, "graph_failures")) { *value = m->graph_failures; return 0; }
    mc_set_error("mc_llm_get_option: unknown option '%s'", name);
    return 1;
}

extern "C" int mc_llm_profile_kinds(void) { return PK_COUNT; }

// total_ms / launches: [PK_COUNT] each, for phase 0 (prefill) or 1 (decode), over the launches since the "profile" option was last enabled
extern "C" int mc_llm_profile_read(void* handle, int phase, double* total_ms, int64_t* launches) {
    Llm* m = (Llm*)handle;
    if (!m || !total_ms || !launches || (phase != 0 && phase != 1)) { mc_set_error("mc_llm_profile_read: bad arguments"); return 1; }
    for (int k = 0; k < PK_COUNT; ++k) { total_ms[k] = 0.0; launches[k] = 0; }
    for (auto& r : m->prof) {
        if (r.phase != phase) continue;
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) { mc_set_error("mc_llm_profile_read: %s", hipGetErrorString(e)); return 2; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        total_ms[r.kind] += t; launches[r.kind] += 1;
    }
    return 0;
}

extern "C" int mc_llm_set_key_mask(void* handle, const void* key_valid, int64_t row_stride) {
    Llm* m = (Llm*)handle;
    if (!m || (key_valid && row_stride <= 0)) { mc_set_error("mc_llm_set_key_mask: bad arguments"); return 1; }
    m->key_mask = key_valid; m->key_mask_stride = key_valid ? row_stride : 0;
    return 0;
}

extern "C" int mc_llm_set_capture(void* handle, void* hidden_snapshots, void* q_snapshots) {
    Llm* m = (Llm*)handle;
    if (!m) { mc_set_error("mc_llm_set_capture: null handle"); return 1; }
    m->cap_hidden = hidden_snapshots; m->cap_q = q_snapshots;
    return 0;
}

extern "C" int mc_llm_set_sampling(void* handle, int do_sample, float temperature, int top_k, float top_p) {
    Llm* m = (Llm*)handle;
    if (!m) { mc_set_error("mc_llm_set_sampling: null handle"); return 1; }
    if (do_sample && (!(temperature > 0.f) || top_k < 0 || !(top_p >= 0.f && top_p <= 1.f))) {
        mc_set_error("mc_llm_set_sampling: bad parameters (temperature=%g top_k=%d top_p=%g)", (double)temperature, top_k, (double)top_p);
        return 1;
    }
    m->do_sample = do_sample != 0; m->temperature = temperature; m->top_k = top_k; m->top_p = top_p;
    return 0;
}

extern "C" int mc_llm_workspace_bytes(void* handle, int M, int B, int Lq, int64_t* bytes) {
    Llm* m = (Llm*)handle;
    if (!m || !bytes || M <= 0 || B <= 0 || Lq <= 0) { mc_set_error("mc_llm_workspace_bytes: bad arguments"); return 1; }
    // the prefill's carve-up and the decode steps' (padded rows) share the buffer; the attention partials follow either
    const Ws w = carve(m->cfg, M, B, Lq, nullptr, false), wd = carve(m->cfg, B, B, 1, nullptr, true);
    *bytes = (int64_t)(w.total > wd.total ? w.total : wd.total) + attn_ws_bytes(m->cfg, B) + 256;
    return 0;
}

extern "C" int mc_llm_prefill(void* handle, void* x_routed, int M, int n_groups, const int32_t* group_start,
                              const int32_t* group_adapter, const int32_t* row_b, const int32_t* row_pos, const int32_t* row_t,
                              const int32_t* out_map, const int32_t* kv_lens, const int32_t* last_rows, int B, int Lq,
                              void* k_cache, void* v_cache, int Smax, void* workspace, void* hidden_out, float* logits_out,
                              int64_t* next_ids, void* stream) {
    Llm* m = (Llm*)handle;
    RUN(check_handle(m, "mc_llm_prefill"));
    if (!x_routed || !group_start || !group_adapter || !row_b || !row_pos || !row_t || !out_map || !k_cache || !v_cache || !workspace ||
        M <= 0 || B <= 0 || Lq <= 0 || n_groups <= 0 || n_groups > kMaxGroups || Smax < Lq) {
        mc_set_error("mc_llm_prefill: bad arguments (M=%d B=%d Lq=%d Smax=%d groups=%d, at most %d groups)", M, B, Lq, Smax, n_groups,
                     kMaxGroups);
        return 1;
    }
    for (int g = 0; g < n_groups; ++g)
        if (group_adapter[g] < 0 || group_adapter[g] >= m->cfg.n_adapters || group_start[g + 1] < group_start[g] || group_start[g + 1] > M) {
            mc_set_error("mc_llm_prefill: bad group %d (adapter %d rows %d..%d)", g, group_adapter[g], group_start[g], group_start[g + 1]);
            return 1;
        }
    // one-shot key mask: this call's, then gone (also on the error paths below: the handle field is already cleared)
    m->call_key_mask = m->key_mask; m->call_key_mask_stride = m->key_mask_stride;
    m->key_mask = nullptr; m->key_mask_stride = 0;
    struct MaskGuard { Llm* m; ~MaskGuard() { m->call_key_mask = nullptr; m->call_key_mask_stride = 0; } } mask_guard{m};
    const mc_llm_config& c = m->cfg;
    // one-shot capture buffers: this call's, then gone (also on the error paths below)
    char* cap_h = (char*)m->cap_hidden; char* cap_q = (char*)m->cap_q;
    m->cap_hidden = nullptr; m->cap_q = nullptr;
    const size_t snap_h = (size_t)M * c.hidden * 2, snap_q = (size_t)B * Lq * c.n_heads * c.head_dim * 2;
    Ws w = carve(c, M, B, Lq, (char*)workspace, false);
    auto snapshot = [&](int l) -> int {               // after layer l (l = -1: the embeddings)
        hipError_t e = hipSuccess;
        if (cap_h) e = hipMemcpyAsync(cap_h + (size_t)(l + 1) * snap_h, x_routed, snap_h, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e == hipSuccess && cap_q && l >= 0) e = hipMemcpyAsync(cap_q + (size_t)l * snap_q, w.qseq, snap_q, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e != hipSuccess) { mc_set_error("mc_llm_prefill: capture copy: %s", hipGetErrorString(e)); return 2; }
        return 0;
    };
    RUN(snapshot(-1));
    RUN(mc_rms_scale_bf16(x_routed, c.hidden, w.rs, M, c.hidden, c.rms_eps, stream));
    // generate() only reads the last token's hidden state of the LAST layer (lm_head on row -1, multimodal_llama.py:720 + greedy_search):
    // that layer still projects q|k|v for every row (the cache needs all keys), but its attention, o_proj and MLP run for the B last
    // tokens only - as a decode step would, reading the keys the projection just stored.  Needs all last tokens on one adapter.
    const bool tail = m->tail_adapter >= 0 && !hidden_out && last_rows && kv_lens && (logits_out || next_ids) && B <= 512 && !cap_h && !cap_q;
    const int tail_adapter = m->tail_adapter;
    // ONE-SHOT: the option is the caller's promise that every last_rows entry of THIS batch is routed to that adapter (last_rows lives on
    // the device; checking it here would cost a synchronising copy per prefill).  It is consumed by the call, so a promise made for one
    // batch can never be applied silently to the next: a caller that wants the tail path sets it before every prefill.
    m->tail_adapter = -1;
    const int full_layers = tail ? c.n_layers - 1 : c.n_layers;
    for (int l = 0; l < full_layers; ++l) {
        RUN(layer_forward(m, l, (char*)x_routed, c.hidden, M, n_groups, group_start, group_adapter, w, row_b, row_pos, row_t, out_map, kv_lens, B,
                          Lq, (char*)k_cache, (char*)v_cache, Smax, false, 1, nullptr, stream));
        if (cap_h || cap_q) RUN(snapshot(l));
    }
    if (tail) {
        const int l = c.n_layers - 1;
        const int64_t hd = c.hidden, D = c.head_dim, H = c.n_heads, Hkv = c.n_kv_heads;
        RUN(layer_forward(m, l, (char*)x_routed, c.hidden, M, n_groups, group_start, group_adapter, w, row_b, row_pos, row_t, out_map, kv_lens, B,
                          Lq, (char*)k_cache, (char*)v_cache, Smax, false, 1, nullptr, stream, 1));
        // last tokens: residual rows, rotated queries (sequence order: row b Lq + len_b - 1 of qseq), attention over the stored keys
        RUN(mc_copy_rows_bf16(x_routed, hd, last_rows, w.xl, w.ldx, nullptr, B, (int)hd, stream));
        RUN(mc_gather_last_rows_bf16(w.qseq, H * D, kv_lens, Lq, w.qkv, H * D, B, (int)(H * D), stream));
        const size_t kv_layer = (size_t)B * Hkv * Smax * D * 2;
        char* kcl = (char*)k_cache + (size_t)l * kv_layer;
        char* vcl = (char*)v_cache + (size_t)l * kv_layer;
        const int nsplit = decode_nsplit(c, B, Smax);
        const mc_attn_mask kmask{m->call_key_mask, m->call_key_mask_stride, 0, 0};
        const mc_attn_mask* mk_tail = m->call_key_mask ? &kmask : nullptr;
        RUNP(m, 0, PK_ATTN, stream, mc_attn_decode_bf16(w.qkv, H * D, D, kcl, Hkv * Smax * D, D, (int64_t)Smax * D, vcl, Hkv * Smax * D, D,
                                                        (int64_t)Smax * D, w.attn, w.lda, (char*)workspace + w.total, kv_lens, B, (int)H, (int)Hkv, Smax,
                                                        (int)D, nsplit, 1.0f / sqrtf((float)D), mk_tail, stream));
        const int32_t gs1[2] = {0, B};
        const int32_t ga1[1] = {tail_adapter};
        RUN(layer_forward(m, l, w.xl, w.ldx, B, 1, gs1, ga1, w, nullptr, nullptr, nullptr, nullptr, kv_lens, B, 1, (char*)k_cache, (char*)v_cache, Smax,
                          true, 1, nullptr, stream, 2, 0));
        float* lg = logits_out ? logits_out : (float*)w.logits;
        RUN(head_forward(m, w.xl, B, w, lg, stream, 0));
        if (next_ids) RUN(mc_argmax_step_f32(lg, c.vocab, next_ids, nullptr, 0, nullptr, B, c.vocab, stream));
        return 0;
    }
    if (hidden_out)   // final norm over all rows (forward() API: logits for every position, :720)
        RUN(mc_rmsnorm_bf16(x_routed, c.hidden, m->final_norm, hidden_out, c.hidden, M, c.hidden, c.rms_eps, stream));
    if (last_rows && (logits_out || next_ids)) {
        RUN(mc_copy_rows_bf16(x_routed, c.hidden, last_rows, w.xl, w.ldx, nullptr, B, c.hidden, stream));
        float* lg = logits_out ? logits_out : (float*)w.logits;
        RUN(head_forward(m, w.xl, B, w, lg, stream, 0));
        if (next_ids) RUN(mc_argmax_step_f32(lg, c.vocab, next_ids, nullptr, 0, nullptr, B, c.vocab, stream));
    }
    return 0;
}

static int decode_one_step(Llm* m, int B, int64_t* next_ids, int64_t* out_ids, int64_t ld_out, int32_t* state, void* k_cache,
                           void* v_cache, int Smax, const Ws& w, void* attn_ws, int nsplit, float* logits_step, void* stream) {
    const mc_llm_config& c = m->cfg;
    static const int32_t gstart[2] = {0, 0};
    int32_t gs[2] = {0, B};
    (void)gstart;
    const int32_t gad[1] = {0};
    const int32_t* pos = state;
    const int32_t* kvlen = state + B;
    const int32_t* iota = state + 2 * B;
    const int32_t* zeros = state + 3 * B;
    const int32_t* step = state + 4 * B;
    RUNP(m, 1, PK_OTHER, stream, mc_embed_rows_bf16(m->embed, c.hidden, next_ids, w.xl, w.ldx, nullptr, B, c.hidden, stream));
    for (int l = 0; l < c.n_layers; ++l)
        RUN(layer_forward(m, l, w.xl, w.ldx, B, 1, gs, gad, w, iota, pos, zeros, nullptr, kvlen, B, 1, (char*)k_cache, (char*)v_cache, Smax, true,
                          nsplit, attn_ws, stream));
    float* lg = logits_step ? logits_step : (float*)w.logits;
    RUN(head_forward(m, w.xl, B, w, lg, stream, 1));
    if (m->do_sample)
        RUN(mc_sample_step_f32(lg, c.vocab, next_ids, out_ids, ld_out, step, 0, (const uint32_t*)(state + 4 * B + 1), 0ull, B, c.vocab,
                               m->temperature, m->top_k, m->top_p, nullptr, nullptr, 0, stream));
    else
        RUNP(m, 1, PK_OTHER, stream, mc_argmax_step_f32(lg, c.vocab, next_ids, out_ids, ld_out, step, B, c.vocab, stream));
    RUNP(m, 1, PK_OTHER, stream, mc_decode_state_advance(state, B, stream));
    return 0;
}

// Greedy decode of n_steps tokens entirely on the device.  next_ids [B] holds the token fed first (the prefill's
// argmax) and is updated in place; out_ids[b*ld_out + step] receives each new token, step read from the device state.
// logits_out (optional) [n_steps][B][vocab] fp32 for parity tests (disables graph replay).
extern "C" int mc_llm_decode(void* handle, int B, int n_steps, int64_t* next_ids, int64_t* out_ids, int64_t ld_out, int32_t* state,
                             void* k_cache, void* v_cache, int Smax, int kv_len_max, void* workspace, float* logits_out, void* stream) {
    Llm* m = (Llm*)handle;
    RUN(check_handle(m, "mc_llm_decode"));
    if (!next_ids || !state || !k_cache || !v_cache || !workspace || B <= 0 || n_steps < 0) {
        mc_set_error("mc_llm_decode: bad arguments");
        return 1;
    }
    // every step appends one key per sequence: the longest sequence must still fit after the last step (attn_decode clamps the
    // length to Smax, so an overrun would silently overwrite the last cache slot instead of failing)
    if (kv_len_max < 0 || (int64_t)kv_len_max + n_steps > Smax) {
        mc_set_error("mc_llm_decode: KV cache overflow: %d cached keys + %d steps > Smax %d", kv_len_max, n_steps, Smax);
        return 1;
    }
    if (Smax > c_max_smax(m->cfg)) {
        mc_set_error("mc_llm_decode: Smax %d exceeds the context the workspace is sized for (max_pos %d + 64)", Smax, m->cfg.max_pos);
        return 1;
    }
    m->graph_active = 0;
    m->call_key_mask = m->key_mask; m->call_key_mask_stride = m->key_mask_stride;
    m->key_mask = nullptr; m->key_mask_stride = 0;
    struct MaskGuard { Llm* m; ~MaskGuard() { m->call_key_mask = nullptr; m->call_key_mask_stride = 0; } } mask_guard{m};
    const mc_llm_config& c = m->cfg;
    Ws w = carve(c, B, B, 1, (char*)workspace, true);
    void* attn_ws = (char*)workspace + w.total;
    const int nsplit = decode_nsplit(c, B, Smax);
    hipStream_t s = (hipStream_t)stream;
    if (m->use_graph && (!logits_out || m->graph_logits) && n_steps > 1 && !m->prof_on && !m->call_key_mask) {
        // graphs cannot be captured on the legacy null stream: run this call's launches on the handle's own stream, after everything the
        // caller has queued on stream 0 (ev_in) and before anything it queues afterwards (ev_out)
        hipStream_t gs = s;
        if (s == nullptr) {
            if (!m->own_stream) {
                if (hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess ||
                    hipEventCreateWithFlags(&m->ev_in, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&m->ev_out, hipEventDisableTiming) != hipSuccess) {
                    mc_set_error("mc_llm_decode: cannot create the decode stream: %s", hipGetErrorString(hipGetLastError()));
                    return 2;
                }
            }
            gs = m->own_stream;
            if (hipEventRecord(m->ev_in, s) != hipSuccess || hipStreamWaitEvent(gs, m->ev_in, 0) != hipSuccess) {
                mc_set_error("mc_llm_decode: stream hand-over: %s", hipGetErrorString(hipGetLastError()));
                return 2;
            }
        }
        auto hand_back = [&]() -> int {
            if (gs == s) return 0;
            if (hipEventRecord(m->ev_out, gs) != hipSuccess || hipStreamWaitEvent(s, m->ev_out, 0) != hipSuccess) {
                mc_set_error("mc_llm_decode: stream hand-back: %s", hipGetErrorString(hipGetLastError()));
                return 2;
            }
            return 0;
        };
        Llm::Key key{};               // zero-initialised incl. padding: compared with memcmp
        key.B = B; key.Smax = Smax; key.kc = k_cache; key.vc = v_cache; key.ws = workspace; key.state = state; key.next_ids = next_ids;
        key.out_ids = out_ids; key.logits = nullptr; key.ld_out = ld_out;
        key.sample = m->do_sample; key.top_k = m->do_sample ? m->top_k : 0;
        key.temp = m->do_sample ? m->temperature : 0.f; key.top_p = m->do_sample ? m->top_p : 0.f;
        int slot = -1;
        for (int i = 0; i < Llm::kGraphs; ++i)
            if (m->graph_exec[i] && memcmp(&key, &m->gkey[i], sizeof(key)) == 0) { slot = i; break; }
        if (slot < 0) {
            slot = m->graph_next;
            m->graph_next = (m->graph_next + 1) % Llm::kGraphs;
            if (m->graph_exec[slot]) { (void)hipGraphExecDestroy(m->graph_exec[slot]); m->graph_exec[slot] = nullptr; }
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                int rc = decode_one_step(m, B, next_ids, out_ids, ld_out, state, k_cache, v_cache, Smax, w, attn_ws, nsplit, nullptr, gs);
                hipError_t e2 = hipStreamEndCapture(gs, &graph);
                if (rc != 0) e = hipErrorUnknown;
                else if (e2 != hipSuccess || !graph) e = e2 != hipSuccess ? e2 : hipErrorUnknown;
                else {
                    e = hipGraphInstantiate(&m->graph_exec[slot], graph, nullptr, nullptr, 0);
                    if (e != hipSuccess) m->graph_exec[slot] = nullptr;
                }
                if (graph) (void)hipGraphDestroy(graph);
            }
            (void)hipGetLastError();
            if (m->graph_exec[slot]) { m->gkey[slot] = key; ++m->graph_captures; }
            else {
                ++m->graph_failures;
                if (!m->capture_warned) {
                    fprintf(stderr, "libmc_hip: mc_llm_decode: decode graph capture failed (%s); falling back to one launch per kernel\n",
                            hipGetErrorString(e));
                    m->capture_warned = true;
                }
            }
        }
        if (m->graph_exec[slot]) {
            for (int i = 0; i < n_steps; ++i) {
                hipError_t e = hipGraphLaunch(m->graph_exec[slot], gs);
                if (e != hipSuccess) { mc_set_error("mc_llm_decode: hipGraphLaunch: %s", hipGetErrorString(e)); return 2; }
                if (logits_out) {          // "graph_logits": the replayed step left its logits in the workspace
                    e = hipMemcpyAsync(logits_out + (size_t)i * B * c.vocab, w.logits, (size_t)B * c.vocab * sizeof(float), hipMemcpyDeviceToDevice, gs);
                    if (e != hipSuccess) { mc_set_error("mc_llm_decode: logits copy: %s", hipGetErrorString(e)); return 2; }
                }
            }
            m->graph_active = 1;
            return hand_back();
        }
        RUN(hand_back());             // nothing was queued on the decode stream: the eager path below runs on the caller's stream
    }
    for (int i = 0; i < n_steps; ++i)
        RUN(decode_one_step(m, B, next_ids, out_ids, ld_out, state, k_cache, v_cache, Smax, w, attn_ws, nsplit,
                            logits_out ? logits_out + (size_t)i * B * c.vocab : nullptr, stream));
    return 0;
}
