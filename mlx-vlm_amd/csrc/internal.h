// Library-internal entry points (hidden visibility: not part of the C ABI of include/vlm_hip.h).
// They are the exported operators plus the hooks the decode engine needs to keep the HBM busy
// between the dependent kernels of a step (csrc/engine.hip, csrc/prefetch.hip).
#pragma once
#include <stddef.h>

#define VLM_INTERNAL __attribute__((visibility("hidden")))

// Pacing word of the weight prefetcher: the decode chain publishes "layer i's attention is done" (value i + 1) and
// "the sampler tail has started" (n_layers + 1) with one relaxed agent-scope store; nothing is ordered by it - it
// only tells the side kernel when to start pulling the next layer's weights towards the Infinity Cache.
struct VlmProgress {
  int* word;   // nullptr: no publication
  int value;
};

VLM_INTERNAL int vlm_attn_decode_paged_ex(const void* q, int ldq, const void* kpool, const void* vpool,
                                          const void* block_table, int max_pages, const void* kv_len, int kv_len_add, int B,
                                          int Hq, int Hkv, int D, float scale, int nsplit, void* part_o, void* part_ml,
                                          void* out, int ldo, VlmProgress prog, void* stream);

VLM_INTERNAL int vlm_attn_decode_paged_split_ex(const void* q, int ldq, const void* kpool, const void* vpool,
                                                const void* block_table, int max_pages, const void* kv_len, int kv_len_add,
                                                int B, int Hq, int Hkv, int D, float scale, int nsplit, void* part_o,
                                                void* part_ml, void* tickets, void* out, int ldo, const void* const* touch_ptr,
                                                const size_t* touch_bytes, int n_touch, VlmProgress prog, void* stream);

VLM_INTERNAL int vlm_sample_ex(const void* logits, int ld, int B, int V, void* logprobs, void* scratch, int ldlp, void* tok,
                               void* workspace, float temperature, float top_p, float min_p, int top_k, unsigned seed,
                               const void* step_ptr, VlmProgress prog, void* stream);

VLM_INTERNAL int vlm_sample_greedy_advance_ex(const void* logits, int ld, int B, int V, void* logprobs, int ldlp, void* tok,
                                              void* workspace, void* ctx, void* pos, void* out_ring, int ring_len, void* step,
                                              const void* embed, void* h, int D, int ldh, VlmProgress prog, void* stream);

// ---- weight / KV prefetch (csrc/prefetch.hip)
struct VlmPfSeg {
  const void* p;
  size_t bytes;
};
// what one decoder layer (or the lm_head chunk) streams: weight segments + that layer's K / V pools
struct VlmPfItem {
  VlmPfSeg seg[4];
  int nseg;
  const void* kbase;   // layer's K pool (nullptr: no KV prefetch for this item)
  const void* vbase;
  int need;            // persistent form: start when *progress >= need; skip when *progress >= need + 2
};
struct VlmPfKv {
  const int* ctx;           // [B] keys in the cache (device)
  const int* block_table;   // [B][max_pages] or nullptr (identity layout)
  int max_pages, B;
  size_t page_bytes;        // bytes of one 64-token page of one pool (all kv heads)
};

VLM_INTERNAL int vlm_prefetch_launch(const VlmPfItem* item, const VlmPfKv* kv, int wgs, void* stream);
VLM_INTERNAL int vlm_prefetch_persistent_launch(const VlmPfItem* items_dev, int n_items, const VlmPfKv* kv, int* progress,
                                                unsigned* exit_count, int wgs, void* stream);

// ---- fused o_proj + gate/up + down of a decoder layer at batch 1 (csrc/mlp_fused.hip)
VLM_INTERNAL int vlm_mlp_fused_supported(int D, int I, int KO);
VLM_INTERNAL int vlm_mlp_fused_launch(const void* attn, void* h, const void* wo, const void* ln2_w, const void* wgu,
                                      const void* wdown, void* g_h, void* g_act, void* epoch, void* err, float eps, int D,
                                      int I, int KO, void* stamps, int mode, void* stream);

// ---- skinny-M decode GEMM on the matrix cores (csrc/gemv_mfma.hip): 3 <= M <= 16 batch rows
struct VlmRopeKv {
  const int* pos;            // [M] rope position (all three M-RoPE axes are equal for a decoded text token)
  const int* slot;           // [M] KV slot (= tokens already in the cache)
  const float* inv_freq;     // [D/2]
  const int* block_table;    // [M][max_pages] or nullptr (identity layout)
  int max_pages, Hq, Hkv, D;
  unsigned short* kpool;     // [page][Hkv][D/8][64][8]
  unsigned short* vpool;     // [page][Hkv][D][64 key slots]
  float qk_scale = 1.f;      // q and k times this, rounded to bf16, before the rotation (SuScaledRoPE, rope_utils.py:174-176)
  int long_from = 0;         // > 0: inv_freq = [2][D/2] (short, long); long for the whole step when any row's slot >= long_from
};
// split-K workspace of the bf16 GEMMs for kernels captured on another stream than the one that replays them (gemm_bf16.hip)
VLM_INTERNAL int vlm_gemm_splitk_share(void* from_stream, void* to_stream);
VLM_INTERNAL void vlm_gemm_splitk_unshare(void* to_stream);
VLM_INTERNAL size_t vlm_gemv_mfma_ws_bytes(void);
// -> 0 done, > 0 error, -1 shape not handled (take the v_dot2c kernels).  ws: zero-initialised workspace of
// vlm_gemv_mfma_ws_bytes() owned by the caller (one per engine: launches that share it must be stream-ordered), or nullptr
// (no K split across workgroups).  rk != nullptr: the qkv projection with M-RoPE + paged KV write.
VLM_INTERNAL int vlm_gemv_mfma_try(const void* x, const void* W, const void* bias, const void* res, const void* norm_w, void* y,
                                   int M, int N, int K, int ldx, int ldw, int ldy, int ldres, float eps, int epilogue,
                                   const VlmRopeKv* rk, void* ws, void* stream);
VLM_INTERNAL int vlm_gemv_mfma_try_w4(const void* x, const void* Wq, const void* Wsb, const void* bias, const void* res,
                                      const void* norm_w, void* y, int M, int N, int K, int ldx, int ldy, int ldres, float eps,
                                      int epilogue, const VlmRopeKv* rk, void* ws, void* stream);
VLM_INTERNAL void vlm_gemv_set_variant(int bits);   // A/B bits of the batch-1 launch shapes (process-wide; VLM_TUNE_GEMV_VARIANT)
VLM_INTERNAL int vlm_gemv_bf16_ex(const void* x, const void* W, const void* bias, const void* res, const void* norm_w, void* y,
                                  int M, int N, int K, int ldx, int ldw, int ldy, int ldres, float eps, int epilogue, int mfma,
                                  void* ws, void* stream);
VLM_INTERNAL int vlm_gemv_qkv_rope_kvwrite_ex(const void* h, const void* norm_w, float eps, const void* Wqkv, const void* bqkv,
                                              void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D, const void* pos,
                                              const void* slot, const void* inv_freq, const void* block_table, int max_pages,
                                              void* kpool, void* vpool, int mfma, void* ws, float qk_scale, int long_from,
                                              void* stream);
VLM_INTERNAL int vlm_gemv_w4_ex(const void* x, const void* Wq, const void* Wsb, const void* bias, const void* res, const void* norm_w,
                                void* y, int M, int N, int K, int ldx, int ldy, int ldres, float eps, int epilogue, int mfma, void* ws,
                                void* stream);
VLM_INTERNAL int vlm_gemv_w4_qkv_rope_kvwrite_ex(const void* h, const void* norm_w, float eps, const void* Wq, const void* Wsb,
                                                 const void* bqkv, void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D,
                                                 const void* pos, const void* slot, const void* inv_freq, const void* block_table,
                                                 int max_pages, void* kpool, void* vpool, int mfma, void* ws, float qk_scale,
                                                 int long_from, void* stream);

