// Native layer loops for the Qwen2-VL hot path on MI355X: ViT forward, LLM prefill
// and the per-token decode step (optionally captured in a hipGraph), composed from
// the operator entry points of this library.
//
// Mirrors (behaviour, not code):
//   VisionModel.__call__ / Qwen2VLVisionBlock / PatchMerger
//       reference mlx_vlm/models/qwen2_vl/vision.py:105-120,177-194,257-290
//   Qwen2Model / Qwen2VLDecoderLayer / Attention / LanguageModel logits
//       reference mlx_vlm/models/qwen2_vl/language.py:66-120,136-154,170-200,514-517
//   the decode loop body of generate_step (reference mlx_vlm/generate/ar.py:334-389,498-508)
// The reference keeps these loops in Python over MLX's lazy graph; here one call
// enqueues the whole chain on a HIP stream with no host work in between.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <new>
#include <vector>

#include "../../include/vlm_hip.h"
#include "internal.h"

#define TRY(expr)            \
  do {                       \
    int rc__ = (expr);       \
    if (rc__ != 0) return rc__; \
  } while (0)

namespace {

// One captured decode step.  The key is everything the captured launches bake in: the argument block and the
// KV-pool view (both plain pointers / scalars).  A continuous batch changes width (1/2/4/8 rows) as requests come
// and go, and a single-stream generate may interleave with it: each shape keeps its graph instead of re-capturing.
struct DecodeGraph {
  vlm_penalty_args pen;                   // copy of *args.penalties (zeroed when there are none)
  vlm_decode_args args;
  vlm_kv_pool kv;
  hipGraph_t graph;
  hipGraphExec_t exec;
  int launches;
  unsigned long long last_use;
};
constexpr size_t MAX_DECODE_GRAPHS = 16;

// A/B knobs of the captured step (vlm_llm_set_tuning; measured defaults)
struct Tuning {
  int mfma_gemv = 1;
  int attn_pagesplit = 16;                // vlm_attn_decode_paged_split with up to this many workgroups per (row, kv head)
  int gemv_variant = 0;                   // A/B bits of the batch-1 GEMV launch shapes (VLM_TUNE_GEMV_VARIANT)
  int attn_merge = 1;                     // 1: one-row steps merge the page-split partials in the o_proj prologue
};

struct Llm {
  Tuning tune;
  void* mfma_ws = nullptr;                // split-K partial tiles + tickets of the skinny-M decode GEMM (gemv_mfma.hip)
  unsigned* attn_tickets = nullptr;       // [4096] arrival words of the page-split decode attention (zero between launches)
  // bf16 scratch of the prefill GEMMs over 4-bit weights: ONE buffer PER STREAM (an admission prefill on the side stream
  // and a generate_step prefill on the main stream of the same quantized model must not share dequantised weights),
  // each sized once for the largest matrix of the model - never grown, never freed while the engine lives
  struct WScratch { hipStream_t st; void* p; };
  std::vector<WScratch> wscratch;
  vlm_llm_config cfg;
  std::vector<vlm_llm_layer> layers;
  vlm_llm_globals g{};
  vlm_kv_pool kv{};
  std::vector<DecodeGraph> graphs;
  hipGraphExec_t exec = nullptr;      // the graph selected by the last vlm_llm_decode_graph_build
  unsigned long long tick = 0;
  int launches = 0;
};

inline bool same_pen(const vlm_penalty_args& x, const vlm_penalty_args* y) {
  vlm_penalty_args z{};
  if (!y) y = &z;
  return x.hist == y->hist && x.hist_len == y->hist_len && x.hist_cap == y->hist_cap && x.rep_penalty == y->rep_penalty &&
         x.rep_ctx == y->rep_ctx && x.pres_penalty == y->pres_penalty && x.pres_ctx == y->pres_ctx &&
         x.freq_penalty == y->freq_penalty && x.freq_ctx == y->freq_ctx && x.bias_idx == y->bias_idx &&
         x.bias_val == y->bias_val && x.n_bias == y->n_bias && x.row_params == y->row_params && x.bias_stride == y->bias_stride;
}

inline bool same_key(const DecodeGraph& g, const vlm_decode_args& a, const vlm_kv_pool& kv) {
  const vlm_decode_args& b = g.args;
  return same_pen(g.pen, a.penalties) && a.B == b.B && a.tok == b.tok && a.pos == b.pos && a.ctx == b.ctx && a.step == b.step && a.h == b.h &&
         a.qkv == b.qkv && a.attn == b.attn && a.act == b.act && a.logits == b.logits && a.logprobs == b.logprobs &&
         a.scratch == b.scratch && a.part_o == b.part_o && a.part_ml == b.part_ml && a.sample_ws == b.sample_ws &&
         a.out_ring == b.out_ring && a.ring_len == b.ring_len && a.nsplit == b.nsplit && a.temperature == b.temperature &&
         a.top_p == b.top_p && a.min_p == b.min_p && a.top_k == b.top_k && a.seed == b.seed && a.flags == b.flags &&
         kv.kpool == g.kv.kpool &&
         kv.vpool == g.kv.vpool && kv.layer_stride == g.kv.layer_stride && kv.block_table == g.kv.block_table &&
         kv.max_pages == g.kv.max_pages && kv.kpool8 == g.kv.kpool8 && kv.vpool8 == g.kv.vpool8 && kv.ksb == g.kv.ksb &&
         kv.vsb == g.kv.vsb && kv.q8_skip_last == g.kv.q8_skip_last;
}

inline void drop_graph(DecodeGraph& g) {
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
}

struct Vit {
  vlm_vit_config cfg;
  std::vector<vlm_vit_block> blocks;
  vlm_vit_globals g{};
};

inline char* off(void* p, size_t bytes) { return static_cast<char*>(p) + bytes; }

}  // namespace

extern "C" int vlm_abi_version(void) { return 8; }

// ------------------------------------------------------------------ LLM
extern "C" int vlm_llm_create(const vlm_llm_config* cfg, void** handle) {
  if (!cfg || !handle) return 1;
  if (cfg->hidden <= 0 || cfg->n_layers <= 0 || cfg->n_heads <= 0 || cfg->n_kv_heads <= 0 || cfg->head_dim <= 0) return 1;
  Llm* m = new (std::nothrow) Llm();
  if (!m) return 1;
  m->cfg = *cfg;
  m->layers.resize(cfg->n_layers);
  // (allocated here, not lazily: the first decode step may already run inside a stream capture.  No device - host-side
  //  construction in the CPU tests - leaves it null: the split-K forms are then simply not taken)
  if (hipMalloc(&m->mfma_ws, vlm_gemv_mfma_ws_bytes()) != hipSuccess) {
    m->mfma_ws = nullptr;
    (void)hipGetLastError();
  } else if (hipMemset(m->mfma_ws, 0, vlm_gemv_mfma_ws_bytes()) != hipSuccess) {
    (void)hipFree(m->mfma_ws);
    delete m;
    return 1012;
  }
  if (hipMalloc((void**)&m->attn_tickets, 4096 * sizeof(unsigned)) != hipSuccess) {
    m->attn_tickets = nullptr;            // (no device: the page-split form is then not taken)
    (void)hipGetLastError();
  } else if (hipMemset(m->attn_tickets, 0, 4096 * sizeof(unsigned)) != hipSuccess) {
    return 1013;
  }
  *handle = m;
  return 0;
}

extern "C" int vlm_llm_destroy(void* handle) {
  Llm* m = static_cast<Llm*>(handle);
  if (!m) return 1;
  for (DecodeGraph& g : m->graphs) drop_graph(g);
  for (auto& ws : m->wscratch) (void)hipFree(ws.p);
  if (m->mfma_ws) (void)hipFree(m->mfma_ws);
  if (m->attn_tickets) (void)hipFree(m->attn_tickets);
  delete m;
  return 0;
}

extern "C" int vlm_llm_set_tuning(void* handle, int key, int value) {
  Llm* m = static_cast<Llm*>(handle);
  if (!m) return 1;
  int* slot = nullptr;
  switch (key) {
    case VLM_TUNE_MFMA_GEMV: if (value < 0 || value > 1) return 1; slot = &m->tune.mfma_gemv; break;
    case VLM_TUNE_ATTN_PAGESPLIT: if (value < 0 || value > 32) return 1; slot = &m->tune.attn_pagesplit; break;
    case VLM_TUNE_GEMV_VARIANT: if (value < 0) return 1; slot = &m->tune.gemv_variant; break;
    case VLM_TUNE_ATTN_MERGE: if (value < 0 || value > 1) return 1; slot = &m->tune.attn_merge; break;
    default: return 1;
  }
  if (key == VLM_TUNE_GEMV_VARIANT) vlm_gemv_set_variant(value);
  if (*slot == value) return 0;
  *slot = value;
  for (DecodeGraph& g : m->graphs) drop_graph(g);     // the captured steps bake the tuning in
  m->graphs.clear();
  m->exec = nullptr;
  return 0;
}

extern "C" int vlm_llm_get_tuning(void* handle, int key) {
  Llm* m = static_cast<Llm*>(handle);
  if (!m) return -1;
  switch (key) {
    case VLM_TUNE_MFMA_GEMV: return m->tune.mfma_gemv;
    case VLM_TUNE_ATTN_PAGESPLIT: return m->tune.attn_pagesplit;
    case VLM_TUNE_GEMV_VARIANT: return m->tune.gemv_variant;
    case VLM_TUNE_ATTN_MERGE: return m->tune.attn_merge;
    default: return -1;
  }
}

extern "C" int vlm_llm_set_layer(void* handle, int layer, const vlm_llm_layer* w) {
  Llm* m = static_cast<Llm*>(handle);
  if (!m || !w || layer < 0 || layer >= m->cfg.n_layers) return 1;
  m->layers[layer] = *w;
  return 0;
}

extern "C" int vlm_llm_set_globals(void* handle, const vlm_llm_globals* g) {
  Llm* m = static_cast<Llm*>(handle);
  if (!m || !g) return 1;
  m->g = *g;
  return 0;
}

extern "C" int vlm_llm_set_kv(void* handle, const vlm_kv_pool* kv) {
  Llm* m = static_cast<Llm*>(handle);
  if (!m || !kv) return 1;
  m->kv = *kv;
  return 0;
}

// ---- one projection, bf16 or MLX 4-bit weights
// prefill: C = epi(A . W^T): 4-bit weights are materialised as bf16 once per projection (vlm_dequant_w4 -> scratch)
static int lin_gemm(Llm* m, const void* A, const void* W, const void* Wsb, const void* bias, const void* res, void* C, int M,
                    int N, int K, int ldc, int ldres, int epi, void* stream) {
  // MLX 4-bit weights: up to 2048 rows the dequant-fused GEMM (vlm_gemm_w4: the packed weights are read once, 4.5 bits
  // each); beyond that the matrix is dequantised once into the stream's scratch and the phased 256x256 bf16 kernel runs
  // (1.4 vs 0.9 PFLOP/s: at that many rows the GEMM time, not the weight traffic, decides).  Same values either way.
  if (Wsb && M <= 2048 && K % 64 == 0)
    return vlm_gemm_w4(A, W, Wsb, bias, res, C, M, N, K, K, ldc, ldres, epi, stream);
  if (Wsb) {
    const vlm_llm_config& c = m->cfg;
    const size_t D = (size_t)c.hidden, KO = (size_t)c.n_heads * c.head_dim, QKV = (size_t)(c.n_heads + 2 * c.n_kv_heads) * c.head_dim;
    size_t cap = QKV * D;
    for (size_t v : {D * KO, 2 * (size_t)c.inter * D, (size_t)c.inter * D, (size_t)((c.vocab + 7) & ~7) * D}) cap = v > cap ? v : cap;
    if ((size_t)N * K > cap) return 1011;
    void* buf = nullptr;
    for (auto& ws : m->wscratch) if (ws.st == (hipStream_t)stream) buf = ws.p;
    if (!buf) {
      if (hipMalloc(&buf, cap * 2) != hipSuccess) return 1011;
      m->wscratch.push_back({(hipStream_t)stream, buf});
    }
    TRY(vlm_dequant_w4(W, Wsb, nullptr, buf, N, K, K, N, stream));
    W = buf;
  }
  return vlm_gemm_bf16(A, W, bias, res, C, M, N, K, K, K, ldc, ldres, epi, stream);
}
// lin_gemm whose split-K reduce launch may carry the caller's next launch (VlmGemmTail, internal.h): *done = 1 when it did
static int lin_gemm_tail(Llm* m, const void* A, const void* W, const void* Wsb, const void* bias, const void* res, void* C, int M,
                         int N, int K, int ldc, int ldres, int epi, const VlmGemmTail* tail, int* done, void* stream) {
  static const bool off = [] { const char* e = getenv("VLM_WIDE_TAILS"); return e && atoi(e) == 0; }();   // A/B knob
  *done = 0;
  if (off) return lin_gemm(m, A, W, Wsb, bias, res, C, M, N, K, ldc, ldres, epi, stream);
  if (Wsb && M <= 2048 && K % 64 == 0)
    return vlm_gemm_w4_tail(A, W, Wsb, bias, res, C, M, N, K, K, ldc, ldres, epi, tail, done, stream);
  if (Wsb) return lin_gemm(m, A, W, Wsb, bias, res, C, M, N, K, ldc, ldres, epi, stream);
  return vlm_gemm_bf16_tail(A, W, bias, res, C, M, N, K, K, K, ldc, ldres, epi, tail, done, stream);
}
// decode: y = epi(x . W^T) for B rows
static int lin_gemv(Llm* m, const void* x, const void* W, const void* Wsb, const void* bias, const void* res, const void* norm_w,
                    void* y, int B, int N, int K, int ldy, int ldres, float eps, int epi, void* stream) {
  if (Wsb) return vlm_gemv_w4_ex(x, W, Wsb, bias, res, norm_w, y, B, N, K, K, ldy, ldres, eps, epi, m->tune.mfma_gemv, m->mfma_ws, stream);
  // B >= 3: batch rows on the matrix cores (gemv_mfma.hip), K split over workgroups through the engine's workspace
  return vlm_gemv_bf16_ex(x, W, bias, res, norm_w, y, B, N, K, K, K, ldy, ldres, eps, epi, m->tune.mfma_gemv, m->mfma_ws, stream);
}

extern "C" int vlm_llm_prefill(void* handle, const vlm_prefill_args* a, void* stream) {
  Llm* m = static_cast<Llm*>(handle);
  if (!m || !a || !a->h || a->T <= 0) return 1;
  const vlm_llm_config& c = m->cfg;
  const int D = c.hidden, hd = c.head_dim, Hq = c.n_heads, Hkv = c.n_kv_heads;
  const int QKV = (Hq + 2 * Hkv) * hd, T = a->T;
  const float scale = c.attn_scale > 0.f ? c.attn_scale : 1.0f / sqrtf((float)hd);
  const float qk_scale = c.rope_qk_scale > 0.f ? c.rope_qk_scale : 1.f;
  bool xn_ready = false;      // xn already holds RMSNorm(h) of the coming layer (the previous down GEMM's split-K reduce launch wrote it)
  for (int i = 0; i < c.n_layers; ++i) {
    const vlm_llm_layer& w = m->layers[i];
    // xn = RMSNorm(h)
    if (!xn_ready) TRY(vlm_rmsnorm_residual(a->h, nullptr, w.ln1_w, a->xn, nullptr, T, D, c.rms_eps, stream));
    // qkv = xn Wqkv^T + b
    TRY(lin_gemm(m, a->xn, w.wqkv, w.wqkv_sb, w.bqkv, nullptr, a->qkv, T, QKV, D, QKV, 0, VLM_EPI_BIAS, stream));
    // M-RoPE on q, k in place + paged KV write
    void* kp = m->kv.kpool ? off(m->kv.kpool, (size_t)i * m->kv.layer_stride * 2) : nullptr;
    void* vp = m->kv.vpool ? off(m->kv.vpool, (size_t)i * m->kv.layer_stride * 2) : nullptr;
    // SuScaledRoPE: the caller names the regime of THIS call (rope_utils.py:168-172: long iff max(cache offset) + tokens of
    // the call > original_max); the long table sits behind the short one
    const void* inv_freq = (a->rope_long && c.rope_long_from > 0) ? off(const_cast<void*>(m->g.inv_freq), (size_t)(hd / 2) * sizeof(float))
                                                                : m->g.inv_freq;
    TRY(vlm_mrope_kvwrite_scaled(a->qkv, QKV, T, Hq, Hkv, hd, a->pos_t, a->pos_h, a->pos_w, inv_freq, c.mrope_sec0,
                                 c.mrope_sec1, a->kv_seq, a->kv_slot, m->kv.block_table, m->kv.max_pages, kp, vp, qk_scale,
                                 stream));
    // causal flash attention over each sequence
    TRY(vlm_attn_prefill(a->qkv, off(a->qkv, (size_t)Hq * hd * 2), off(a->qkv, (size_t)(Hq + Hkv) * hd * 2), a->attn, QKV,
                         QKV, QKV, Hq * hd, a->cu_seqlens, a->nseg, a->total_qblocks, Hq, Hkv, hd, scale, 1, stream));
    // h = h + attn Wo^T
    TRY(lin_gemm(m, a->attn, w.wo, w.wo_sb, nullptr, a->h, a->h, T, D, Hq * hd, D, D, VLM_EPI_RESIDUAL, stream));
    // xn = RMSNorm(h); act = swiglu(xn Wgu^T); h = h + act Wdown^T
    TRY(vlm_rmsnorm_residual(a->h, nullptr, w.ln2_w, a->xn, nullptr, T, D, c.rms_eps, stream));
    TRY(lin_gemm(m, a->xn, w.wgu, w.wgu_sb, nullptr, nullptr, a->act, T, 2 * c.inter, D, c.inter, 0, VLM_EPI_SWIGLU, stream));
    // (round 6) a split-K down GEMM's reduce launch also writes the NEXT layer's RMSNorm(h) (VlmGemmTail: the wide decode steps'
    // mechanism, same bits as the separate launch; the last layer's rows go through the gather below instead)
    int done = 0;
    if (i + 1 < c.n_layers) {
      VlmGemmTail nt{};
      nt.kind = VLM_TAIL_RMSNORM;
      nt.norm_w = m->layers[i + 1].ln1_w; nt.eps = c.rms_eps; nt.xn = a->xn; nt.ldxn = D;
      TRY(lin_gemm_tail(m, a->act, w.wdown, w.wdown_sb, nullptr, a->h, a->h, T, D, c.inter, D, D, VLM_EPI_RESIDUAL, &nt, &done, stream));
    } else {
      TRY(lin_gemm(m, a->act, w.wdown, w.wdown_sb, nullptr, a->h, a->h, T, D, c.inter, D, D, VLM_EPI_RESIDUAL, stream));
    }
    xn_ready = done != 0;
  }
  if (a->n_last > 0) {
    if (!a->last_rows || !a->xlast || !a->logits) return 1;
    // gather the wanted rows, final norm, lm_head on those rows only (reference: all L rows, ar.py:358)
    // gather = embed_gather over the residual stream viewed as a table of T rows
    TRY(vlm_embed_gather(a->last_rows, a->h, a->xlast, a->n_last, D, D, T, stream));
    TRY(vlm_rmsnorm_residual(a->xlast, nullptr, m->g.final_norm_w, a->xlast, nullptr, a->n_last, D, c.rms_eps, stream));
    // a vocabulary that is not a multiple of 8 (Idefics2: 32003): logits rows have pitch VL = vocab rounded up, the head
    // matrix VL rows (the loader pads it with zero rows); the samplers read the first `vocab` columns only
    const int VL = (c.vocab + 7) & ~7;
    TRY(lin_gemm(m, a->xlast, m->g.lm_head, m->g.lm_head_sb, nullptr, nullptr, a->logits, a->n_last, VL, D, VL, 0,
                 VLM_EPI_NONE, stream));
  }
  return 0;
}

static int decode_impl(Llm* m, const vlm_decode_args* a, void* stream, int* launches, bool sample) {
  const vlm_llm_config& c = m->cfg;
  const int D = c.hidden, hd = c.head_dim, Hq = c.n_heads, Hkv = c.n_kv_heads, B = a->B, NL = c.n_layers;
  const int QKV = (Hq + 2 * Hkv) * hd;
  const float scale = c.attn_scale > 0.f ? c.attn_scale : 1.0f / sqrtf((float)hd);
  const float qk_scale = c.rope_qk_scale > 0.f ? c.rope_qk_scale : 1.f;
  // VLM_DECODE_FUSED_TAIL: the step starts from h == embed[tok] and its sampler tail leaves the next step's h behind - the greedy
  // tail (vlm_sample_greedy_advance) or, with a temperature, the sampled one (vlm_sample_advance: final pick + advance + gather)
  const bool fused_flag = sample && (a->flags & VLM_DECODE_FUSED_TAIL);
  const bool fused_tail = fused_flag && a->temperature == 0.f;
  const bool fused_sampled = fused_flag && a->temperature > 0.f;
  if ((a->flags & VLM_DECODE_FUSED_TAIL) && !fused_flag) return 1;   // the flag promises h == embed[tok] at entry
  const Tuning& tn = m->tune;
  // WIDE steps (more than 16 rows): beyond one N tile of the skinny-M MFMA GEMM the projections run on the prefill GEMMs,
  // i.e. a layer is the prefill's launch sequence (vlm_llm_prefill above) with the paged decode attention in place of
  // the flash attention: RMSNorm -> qkv GEMM + bias -> M-RoPE + KV write at slot ctx[b] -> attention over the pages ->
  // o_proj GEMM + residual -> RMSNorm -> gate/up GEMM + SwiGLU -> down GEMM + residual.  Measured per 7B layer
  // (profiles/r03_mfma_shapes.txt D): 192.6 us at 32 rows, 210 at 64 - against 120 per 16-row step.  The normalised rows
  // borrow the attention output buffer (free before the qkv GEMM and again after o_proj).  Two-table RoPE models
  // (SuScaledRoPE): the per-call regime (rope_utils.py:168-172) reaches vlm_mrope_kvwrite_scaled as `long_from`, decided on
  // the device from the rows' slots exactly as the fused qkv kernels of the <= 16-row steps do.
  const bool wide = B > 16;
  if (wide && (!m->kv.block_table || Hq * hd < D || B > 64)) return 1;
  void* const xn = a->attn;
  bool xn_ready = false;      // wide steps: xn already holds RMSNorm(h) of the coming layer (the previous down GEMM's reduce launch wrote it)
  int n = 0;
  // h = embed[tok]
  if (!fused_flag) {
    if (m->g.embed_sb) { TRY(vlm_dequant_w4(m->g.embed, m->g.embed_sb, a->tok, a->h, B, D, D, c.vocab, stream)); }
    else { TRY(vlm_embed_gather(a->tok, m->g.embed, a->h, B, D, D, c.vocab, stream)); }
    ++n;
  } else if (m->g.embed_sb) {
    return 1;      // the fused tail gathers bf16 embedding rows
  }
  for (int i = 0; i < NL; ++i) {
    const vlm_llm_layer& w = m->layers[i];
    void* kp = off(m->kv.kpool, (size_t)i * m->kv.layer_stride * 2);
    void* vp = off(m->kv.vpool, (size_t)i * m->kv.layer_stride * 2);
    // [RMSNorm + qkv GEMV + bias + M-RoPE at pos[b] + k/v write at slot ctx[b]] in one launch
    if (wide) {
      // (round 6) the three small GEMMs of a wide layer are split-K: their reduce launches also do the launch that would read
      // the reduced rows straight back - M-RoPE + KV write here, RMSNorm(ln2) after o_proj, the NEXT layer's RMSNorm(ln1) (or
      // the final norm) after down.  Same values (csrc/gemm_bf16.hip); a GEMM that does not take the split-K route reports
      // done = 0 and the follower is launched as before.
      if (!xn_ready) { TRY(vlm_rmsnorm_residual(a->h, nullptr, w.ln1_w, xn, nullptr, B, D, c.rms_eps, stream)); ++n; }
      VlmGemmTail rt{};
      rt.kind = VLM_TAIL_ROPE_KV;
      rt.Hq = Hq; rt.Hkv = Hkv; rt.D = hd;
      rt.pos = (const int*)a->pos; rt.inv_freq = (const float*)m->g.inv_freq; rt.slot = (const int*)a->ctx;
      rt.block_table = (const int*)m->kv.block_table; rt.max_pages = m->kv.max_pages;
      rt.kpool = (unsigned short*)kp; rt.vpool = (unsigned short*)vp; rt.qk_scale = qk_scale; rt.long_from = c.rope_long_from;
      int done = 0;
      TRY(lin_gemm_tail(m, xn, w.wqkv, w.wqkv_sb, w.bqkv, nullptr, a->qkv, B, QKV, D, QKV, 0, w.bqkv ? VLM_EPI_BIAS : VLM_EPI_NONE,
                        &rt, &done, stream)); ++n;
      if (!done) {
        TRY(vlm_mrope_kvwrite_decode(a->qkv, QKV, B, Hq, Hkv, hd, a->pos, m->g.inv_freq, c.mrope_sec0, c.mrope_sec1, a->ctx,
                                     m->kv.block_table, m->kv.max_pages, kp, vp, qk_scale, c.rope_long_from, stream)); ++n;
      }
    } else if (w.wqkv_sb) {
      TRY(vlm_gemv_w4_qkv_rope_kvwrite_ex(a->h, w.ln1_w, c.rms_eps, w.wqkv, w.wqkv_sb, w.bqkv, a->qkv, QKV, B, D, Hq, Hkv, hd, a->pos,
                                          a->ctx, m->g.inv_freq, m->kv.block_table, m->kv.max_pages, kp, vp, m->tune.mfma_gemv,
                                          m->mfma_ws, qk_scale, c.rope_long_from, stream)); ++n;
    } else {
      TRY(vlm_gemv_qkv_rope_kvwrite_ex(a->h, w.ln1_w, c.rms_eps, w.wqkv, w.bqkv, a->qkv, QKV, B, D, Hq, Hkv, hd, a->pos, a->ctx,
                                       m->g.inv_freq, m->kv.block_table, m->kv.max_pages, kp, vp, m->tune.mfma_gemv, m->mfma_ws,
                                       qk_scale, c.rope_long_from, stream)); ++n;
    }
    // attention over the pages (the new token is already in the cache: kv_len = ctx + 1)
    // combine: the split partials are merged by the attention side (combine kernel / last arriver) into a->attn - 4-bit Wo,
    // bf16 Wo wider than the row-wave GEMV's prologue, and EVERY wide step (its o_proj is a GEMM over a->attn)
    const bool combine = wide || w.wo_sb != nullptr || Hq * hd > 3584;
    // few (row, kv head) pairs: the page-split form - one wave per page stride over up to 256 CUs, the last arriver
    // writes the final vector (part_o / part_ml are sized for 32 splits by the caller)
    int psplit = 0;
    if (tn.attn_pagesplit > 0 && m->attn_tickets && B * Hkv <= 64) {
      psplit = a->nsplit > 1 ? 32 : tn.attn_pagesplit;
      if (psplit > 256 / (B * Hkv)) psplit = 256 / (B * Hkv);
      if (psplit < 2) psplit = 0;
    }
    // uniform 8-bit KV cache (vlm_kv_pool.kpool8): the step attends over the 8-bit pools; the launch quantises the new token.
    // q8_skip_last (ABI v5): the reference's BATCH policy keeps the last layer of a stack deeper than 2 on the unquantised
    // cache (models/cache.py:8-21)
    const bool q8 = m->kv.kpool8 != nullptr && !(m->kv.q8_skip_last && NL > 2 && i == NL - 1);
    if (q8) {
      if (!m->attn_tickets || !m->kv.vpool8 || !m->kv.ksb || !m->kv.vsb) return 1;
      psplit = a->nsplit > 1 ? 32 : 16;
      while (psplit > 1 && B * Hkv * psplit > 2048) psplit >>= 1;          // up to 2 one-wave workgroups per SIMD (209 VGPRs)
      if (B * Hkv > 4096) return 1;                                        // (tickets: one word per (row, kv head))
    }
    // ... and for ONE row over bf16 Wo the merge moves into the o_proj prologue: the attention launch ends at its partial
    // stores (no ticket, no last-arriver pass)
    const bool merge_in_oproj = psplit && tn.attn_merge && B == 1 && !w.wo_sb && Hq * hd <= 2048 && psplit <= 16;
    if (q8) {
      const size_t lo = (size_t)i * m->kv.layer_stride;                   // elements == bytes of the u8 pools
      TRY(vlm_attn_decode_paged_q8(a->qkv, QKV, kp, vp, off(m->kv.kpool8, lo), off(m->kv.vpool8, lo),
                                   off(m->kv.ksb, lo / (hd / 2) * 4), off(m->kv.vsb, lo / (hd / 2) * 4), m->kv.block_table,
                                   m->kv.max_pages, a->ctx, 1, B, Hq, Hkv, hd, scale, psplit, a->part_o, a->part_ml, m->attn_tickets,
                                   merge_in_oproj ? nullptr : a->attn, Hq * hd, 1, stream)); ++n;
    } else if (psplit) {
      TRY(vlm_attn_decode_paged_split(a->qkv, QKV, kp, vp, m->kv.block_table, m->kv.max_pages, a->ctx, 1, B, Hq, Hkv, hd, scale,
                                      psplit, a->part_o, a->part_ml, m->attn_tickets, merge_in_oproj ? nullptr : a->attn,
                                      Hq * hd, stream)); ++n;
    } else if (a->nsplit == 1) {
      // short contexts: one workgroup per (sequence, kv head) -> final bf16 vector, plain o_proj GEMV + residual
      TRY(vlm_attn_decode_paged(a->qkv, QKV, kp, vp, m->kv.block_table, m->kv.max_pages, a->ctx, 1, B, Hq, Hkv, hd, scale, 1,
                                a->part_o, a->part_ml, a->attn, Hq * hd, stream)); ++n;
    } else {
      // long contexts: split-K partials, merged in the o_proj GEMV prologue (bf16 Wo up to 3584 columns: the row-wave
      // kernel) or by the combine kernel (4-bit Wo; wider bf16 Wo, e.g. 32 heads x 128; wide steps)
      TRY(vlm_attn_decode_paged(a->qkv, QKV, kp, vp, m->kv.block_table, m->kv.max_pages, a->ctx, 1, B, Hq, Hkv, hd, scale,
                                a->nsplit, a->part_o, a->part_ml, combine ? a->attn : nullptr, combine ? Hq * hd : 0,
                                stream)); ++n;
    }
    if (wide) {
      VlmGemmTail nt{};
      nt.kind = VLM_TAIL_RMSNORM;
      nt.norm_w = w.ln2_w; nt.eps = c.rms_eps; nt.xn = xn; nt.ldxn = D;
      int done = 0;
      TRY(lin_gemm_tail(m, a->attn, w.wo, w.wo_sb, nullptr, a->h, a->h, B, D, Hq * hd, D, D, VLM_EPI_RESIDUAL, &nt, &done, stream)); ++n;
      if (!done) { TRY(vlm_rmsnorm_residual(a->h, nullptr, w.ln2_w, xn, nullptr, B, D, c.rms_eps, stream)); ++n; }
      TRY(lin_gemm(m, xn, w.wgu, w.wgu_sb, nullptr, nullptr, a->act, B, 2 * c.inter, D, c.inter, 0, VLM_EPI_SWIGLU, stream)); ++n;
      nt.norm_w = i + 1 < NL ? m->layers[i + 1].ln1_w : m->g.final_norm_w;
      TRY(lin_gemm_tail(m, a->act, w.wdown, w.wdown_sb, nullptr, a->h, a->h, B, D, c.inter, D, D, VLM_EPI_RESIDUAL, &nt, &done, stream)); ++n;
      xn_ready = done != 0;
      continue;
    }
    if (merge_in_oproj) {
      TRY(vlm_gemv_attn_out_bf16(a->part_o, a->part_ml, psplit, w.wo, a->h, D, D, Hq, hd, stream)); ++n;
    } else if (psplit || a->nsplit == 1 || combine) {
      TRY(lin_gemv(m, a->attn, w.wo, w.wo_sb, nullptr, a->h, nullptr, a->h, B, D, Hq * hd, D, D, 0.f, VLM_EPI_RESIDUAL, stream)); ++n;
    } else {
      TRY(vlm_gemv_attn_out(a->part_o, a->part_ml, a->nsplit, w.wo, a->h, D, B, D, Hq, hd, stream)); ++n;
    }
    // act = swiglu(RMSNorm(h) Wgu^T);  h = h + act Wdown^T.  Batched steps over bf16 weights hand `act` over in the TILED layout
    // when the down projection takes the row-slice form (one workgroup per CU owns N / 256 output rows and their whole K: no K
    // split across workgroups; csrc/gemv_mfma_rows.hip) - the gate/up launch then writes that layout (refused -> the plain pair)
    bool tiled = false;
    if ((a->flags & VLM_DECODE_ACT16) && B >= 5 && tn.mfma_gemv && !w.wgu_sb && !w.wdown_sb && vlm_gemv_mfma_rows_ok(B, D, c.inter) &&
        c.inter % 8 == 0) {
      const int rc = lin_gemv(m, a->h, w.wgu, nullptr, nullptr, nullptr, w.ln2_w, a->act, B, 2 * c.inter, D, c.inter, 0, c.rms_eps,
                              VLM_EPI_SWIGLU | VLM_EPI_Y_TILED, stream);
      if (rc == 0) tiled = true;
      else if (rc != 2) return rc;
    }
    if (!tiled) TRY(lin_gemv(m, a->h, w.wgu, w.wgu_sb, nullptr, nullptr, w.ln2_w, a->act, B, 2 * c.inter, D, c.inter, 0, c.rms_eps, VLM_EPI_SWIGLU, stream));
    ++n;
    TRY(lin_gemv(m, a->act, w.wdown, w.wdown_sb, nullptr, a->h, nullptr, a->h, B, D, c.inter, D, D, 0.f,
                 VLM_EPI_RESIDUAL | (tiled ? VLM_EPI_X_TILED : 0), stream)); ++n;
  }
  // logits = RMSNorm(h) lm_head^T
  const int VL = (c.vocab + 7) & ~7;      // row pitch of logits / logprobs / scratch (see vlm_llm_prefill)
  if (wide) {
    // (the head matrix has VL rows: the loader pads a vocabulary that is not a multiple of 8 with zero rows, as in the prefill)
    if (!xn_ready) { TRY(vlm_rmsnorm_residual(a->h, nullptr, m->g.final_norm_w, xn, nullptr, B, D, c.rms_eps, stream)); ++n; }
    TRY(lin_gemm(m, xn, m->g.lm_head, m->g.lm_head_sb, nullptr, nullptr, a->logits, B, VL, D, VL, 0, VLM_EPI_NONE, stream)); ++n;
  } else {
    TRY(lin_gemv(m, a->h, m->g.lm_head, m->g.lm_head_sb, nullptr, nullptr, m->g.final_norm_w, a->logits, B, c.vocab, D, VL, 0,
                 c.rms_eps, VLM_EPI_NONE, stream)); ++n;
  }
  if (sample && a->penalties) {
    // logits processors (ar.py:360-364): the fed token joins the history, then bias / penalties on the step's logits
    TRY(vlm_apply_logit_penalties(a->logits, VL, B, c.vocab, a->tok, a->penalties, stream)); ++n;
  }
  if (sample) {
    if (fused_tail) {
      TRY(vlm_sample_greedy_advance(a->logits, VL, B, c.vocab, a->logprobs, VL, a->tok, a->sample_ws, a->ctx, a->pos,
                                    a->out_ring, a->ring_len, a->step, m->g.embed, a->h, D, D, stream)); n += 2;
    } else if (fused_sampled) {
      TRY(vlm_sample_advance(a->logits, VL, B, c.vocab, a->logprobs, a->scratch, VL, a->tok, a->sample_ws, a->temperature, a->top_p,
                             a->min_p, a->top_k, a->seed, a->ctx, a->pos, a->out_ring, a->ring_len, a->step, m->g.embed, a->h, D, D,
                             stream));
      n += vlm_sample_last_launches();          // (the split top-p route issues more launches than the one-workgroup filter)
    } else {
      TRY(vlm_sample(a->logits, VL, B, c.vocab, a->logprobs, a->scratch, VL, a->tok, a->sample_ws, a->temperature, a->top_p,
                     a->min_p, a->top_k, a->seed, a->step, stream));
      // (greedy: partials, log-probs + candidates, pick; sampling: + the draw's partials, + the filter kernel(s) when one is on)
      n += vlm_sample_last_launches();
      TRY(vlm_decode_advance(a->ctx, a->pos, a->tok, a->out_ring, a->ring_len, a->step, B, stream)); ++n;
    }
  }
  if (launches) *launches = n;
  return 0;
}

extern "C" int vlm_llm_decode_step(void* handle, const vlm_decode_args* a, void* stream) {
  Llm* m = static_cast<Llm*>(handle);
  if (!m || !a || a->B <= 0 || !m->kv.kpool) return 1;
  return decode_impl(m, a, stream, &m->launches, true);
}

// embeddings -> logits only (the module-contract path: language_model(y, cache=...) at L == 1;
// the caller samples and advances ctx/pos itself)
extern "C" int vlm_llm_decode_forward(void* handle, const vlm_decode_args* a, void* stream) {
  Llm* m = static_cast<Llm*>(handle);
  if (!m || !a || a->B <= 0 || !m->kv.kpool) return 1;
  return decode_impl(m, a, stream, &m->launches, false);
}

extern "C" int vlm_llm_decode_graph_build(void* handle, const vlm_decode_args* a, void* stream) {
  // `stream` is unused for the capture itself: capturing on the (legacy) default stream is illegal, so the
  // step is recorded on a private capture stream (nothing executes during capture) and the instantiated
  // graph can then be launched on any stream.
  (void)stream;
  Llm* m = static_cast<Llm*>(handle);
  if (!m || !a || a->B <= 0 || !m->kv.kpool) return 1;
  for (DecodeGraph& g : m->graphs) {
    if (same_key(g, *a, m->kv)) {
      g.last_use = ++m->tick;
      m->exec = g.exec;
      m->launches = g.launches;
      return 0;
    }
  }
  m->exec = nullptr;
  hipStream_t cap = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&cap, hipStreamNonBlocking);
  if (e != hipSuccess) return 1000 + (int)e;
  DecodeGraph ng{a->penalties ? *a->penalties : vlm_penalty_args{}, *a, m->kv, nullptr, nullptr, 0, ++m->tick};
  // wide steps run split-K GEMMs: the capture stream borrows the launch stream's workspace (the replays are ordered there)
  const bool wide = a->B > 16;
  if (wide && vlm_gemm_splitk_share(stream, (void*)cap) != 0) { (void)hipStreamDestroy(cap); return 1008; }
  e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    if (wide) vlm_gemm_splitk_unshare((void*)cap);
    (void)hipStreamDestroy(cap); return 1000 + (int)e;
  }
  int rc = decode_impl(m, a, (void*)cap, &ng.launches, true);
  e = hipStreamEndCapture(cap, &ng.graph);
  if (wide) vlm_gemm_splitk_unshare((void*)cap);
  (void)hipStreamDestroy(cap);
  if (rc != 0) { drop_graph(ng); return rc; }
  if (e != hipSuccess) { drop_graph(ng); return 1000 + (int)e; }
  e = hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { drop_graph(ng); return 1000 + (int)e; }
  if (m->graphs.size() >= MAX_DECODE_GRAPHS) {      // evict the least recently selected one
    size_t lru = 0;
    for (size_t i = 1; i < m->graphs.size(); ++i)
      if (m->graphs[i].last_use < m->graphs[lru].last_use) lru = i;
    drop_graph(m->graphs[lru]);
    m->graphs.erase(m->graphs.begin() + (long)lru);
  }
  m->graphs.push_back(ng);
  m->exec = ng.exec;
  m->launches = ng.launches;
  return 0;
}

extern "C" int vlm_llm_decode_graph_launch(void* handle, void* stream) {
  Llm* m = static_cast<Llm*>(handle);
  if (!m || !m->exec) return 1;
  hipError_t e = hipGraphLaunch(m->exec, (hipStream_t)stream);
  return e == hipSuccess ? 0 : 1000 + (int)e;
}

extern "C" int vlm_llm_decode_launches(void* handle) {
  Llm* m = static_cast<Llm*>(handle);
  return m ? m->launches : -1;
}

// ------------------------------------------------------------------ ViT
extern "C" int vlm_vit_create(const vlm_vit_config* cfg, void** handle) {
  if (!cfg || !handle || cfg->depth <= 0 || cfg->embed_dim <= 0 || cfg->n_heads <= 0) return 1;
  Vit* v = new (std::nothrow) Vit();
  if (!v) return 1;
  v->cfg = *cfg;
  v->blocks.resize(cfg->depth);
  *handle = v;
  return 0;
}

extern "C" int vlm_vit_destroy(void* handle) {
  Vit* v = static_cast<Vit*>(handle);
  if (!v) return 1;
  delete v;
  return 0;
}

extern "C" int vlm_vit_set_block(void* handle, int i, const vlm_vit_block* w) {
  Vit* v = static_cast<Vit*>(handle);
  if (!v || !w || i < 0 || i >= v->cfg.depth) return 1;
  v->blocks[i] = *w;
  return 0;
}

extern "C" int vlm_vit_set_globals(void* handle, const vlm_vit_globals* g) {
  Vit* v = static_cast<Vit*>(handle);
  if (!v || !g) return 1;
  v->g = *g;
  return 0;
}

extern "C" int vlm_vit_forward(void* handle, const vlm_vit_args* a, void* stream) {
  Vit* v = static_cast<Vit*>(handle);
  if (!v || !a || !a->patches || a->N <= 0) return 1;
  const vlm_vit_config& c = v->cfg;
  const int E = c.embed_dim, H = c.n_heads, hd = E / H, N = a->N, MH = c.mlp_hidden;
  const int mm = c.merge * c.merge;
  if (N % mm != 0) return 2;
  const float scale = 1.0f / sqrtf((float)hd);
  // patch projection (Conv3d with stride == kernel is a GEMM)
  TRY(vlm_gemm_bf16(a->patches, v->g.wpatch, nullptr, nullptr, a->x, N, E, c.patch_k, c.patch_k, c.patch_k, E, 0, VLM_EPI_NONE, stream));
  for (int i = 0; i < c.depth; ++i) {
    const vlm_vit_block& w = v->blocks[i];
    TRY(vlm_layernorm(a->x, w.ln1_w, w.ln1_b, a->xn, N, E, c.ln_eps, stream));
    if (c.qk_interleaved && a->sin_tab == off(const_cast<void*>(a->cos_tab), (size_t)N * (hd / 2) * sizeof(float))) {
      // 2-D rope in the GEMM epilogue (one HBM pass less per block)
      TRY(vlm_gemm_bf16_rope2d(a->xn, w.wqkv, w.bqkv, a->cos_tab, a->qkv, N, 3 * E, E, E, E, 3 * E, hd, 2 * E, stream));
    } else {
      if (c.qk_interleaved) return 1;     // interleaved weights need the fused form
      TRY(vlm_gemm_bf16(a->xn, w.wqkv, w.bqkv, nullptr, a->qkv, N, 3 * E, E, E, E, 3 * E, 0, VLM_EPI_BIAS, stream));
      TRY(vlm_rope2d_vision(a->qkv, a->cos_tab, a->sin_tab, N, H, hd, 3 * E, stream));
    }
    TRY(vlm_attn_prefill(a->qkv, off(a->qkv, (size_t)E * 2), off(a->qkv, (size_t)2 * E * 2), a->attn, 3 * E, 3 * E, 3 * E, E,
                         a->cu_seqlens, a->nseg, a->total_qblocks, H, H, hd, scale, a->uniform_segments ? 2 : 0, stream));
    TRY(vlm_gemm_bf16(a->attn, w.wproj, w.bproj, a->x, a->x, N, E, E, E, E, E, E, VLM_EPI_BIAS | VLM_EPI_RESIDUAL, stream));
    TRY(vlm_layernorm(a->x, w.ln2_w, w.ln2_b, a->xn, N, E, c.ln_eps, stream));
    TRY(vlm_gemm_bf16(a->xn, w.wfc1, w.bfc1, nullptr, a->mlp, N, MH, E, E, E, MH, 0, VLM_EPI_BIAS | VLM_EPI_GELU_FAST, stream));
    TRY(vlm_gemm_bf16(a->mlp, w.wfc2, w.bfc2, a->x, a->x, N, E, MH, MH, MH, E, E, VLM_EPI_BIAS | VLM_EPI_RESIDUAL, stream));
  }
  // PatchMerger: LN -> [N/4, 4E] -> Linear + GELU(erf) -> Linear
  TRY(vlm_layernorm(a->x, v->g.ln_q_w, v->g.ln_q_b, a->xn, N, E, c.ln_eps, stream));
  const int Nm = N / mm, EM = E * mm;
  TRY(vlm_gemm_bf16(a->xn, v->g.wm0, v->g.bm0, nullptr, a->mrg, Nm, EM, EM, EM, EM, EM, 0, VLM_EPI_BIAS | VLM_EPI_GELU_ERF, stream));
  TRY(vlm_gemm_bf16(a->mrg, v->g.wm2, v->g.bm2, nullptr, a->out, Nm, c.out_dim, EM, EM, EM, c.out_dim, 0, VLM_EPI_BIAS, stream));
  return 0;
}

// ------------------------------------------------------------------ generic pre-LN encoder layers (SigLIP / CLIP towers)
// The layer loop of the other model families' vision towers as ONE native call (reference: the Python loops of
// mlx_vlm/models/idefics2/vision.py:141-187, llava_bunny/vision.py:139-200, phi3_v/vision.py:117-175 - EncoderLayer:
// x = x + out_proj(attn(LN1(x))); x = x + fc2(act(fc1(LN2(x))))): n_layers x 7 launches enqueued with no host work in
// between (the Python loops paid ~10 us of host time per launch: Idefics2's 4-image prefill was host-bound).
extern "C" int vlm_encoder_forward(const vlm_enc_layer* layers, int n_layers, void* x, void* xn, void* qkv, void* attn,
                                   void* mlp, int N, int E, int H, int head_dim, int MH, float ln_eps, int act_epilogue,
                                   const void* cu_seqlens, int nseg, int total_qblocks, float scale, int uniform_segments,
                                   void* stream) {
  if (!layers || n_layers < 0 || !x || !xn || !qkv || !attn || !mlp || !cu_seqlens || N <= 0 || E <= 0 || H <= 0 || head_dim <= 0 ||
      MH <= 0)
    return 1;
  if (act_epilogue != VLM_EPI_GELU_FAST && act_epilogue != VLM_EPI_GELU_ERF && act_epilogue != 0) return 1;
  const int HD = H * head_dim;                       // (padded) attention width: qkv rows are [q | k | v] of HD columns each
  for (int i = 0; i < n_layers; ++i) {
    const vlm_enc_layer& w = layers[i];
    TRY(vlm_layernorm(x, w.ln1_w, w.ln1_b, xn, N, E, ln_eps, stream));
    TRY(vlm_gemm_bf16(xn, w.wqkv, w.bqkv, nullptr, qkv, N, 3 * HD, E, E, E, 3 * HD, 0, VLM_EPI_BIAS, stream));
    TRY(vlm_attn_prefill(qkv, off(qkv, (size_t)HD * 2), off(qkv, (size_t)2 * HD * 2), attn, 3 * HD, 3 * HD, 3 * HD, HD, cu_seqlens,
                         nseg, total_qblocks, H, H, head_dim, scale, uniform_segments ? 2 : 0, stream));
    TRY(vlm_gemm_bf16(attn, w.wo, w.bo, x, x, N, E, HD, HD, HD, E, E, VLM_EPI_BIAS | VLM_EPI_RESIDUAL, stream));
    TRY(vlm_layernorm(x, w.ln2_w, w.ln2_b, xn, N, E, ln_eps, stream));
    TRY(vlm_gemm_bf16(xn, w.w1, w.b1, nullptr, mlp, N, MH, E, E, E, MH, 0, VLM_EPI_BIAS | act_epilogue, stream));
    TRY(vlm_gemm_bf16(mlp, w.w2, w.b2, x, x, N, E, MH, MH, MH, E, E, VLM_EPI_BIAS | VLM_EPI_RESIDUAL, stream));
  }
  return 0;
}
