// Library-internal entry points (hidden visibility: not part of the C ABI of include/vlm_hip.h).
// They are the forms of the exported operators that the decode engine (csrc/engine.hip) needs.
#pragma once
#include <stddef.h>

#define VLM_INTERNAL __attribute__((visibility("hidden")))

// decode form of vlm_mrope_kvwrite_scaled (csrc/rope.hip): SuScaledRoPE's per-call regime decided on the device
VLM_INTERNAL int vlm_mrope_kvwrite_decode(void* qkv, int ld, int B, int Hq, int Hkv, int D, const void* pos, const void* inv_freq,
                                          int sec0, int sec1, const void* slot, const void* block_table, int max_pages,
                                          void* kpool, void* vpool, float qk_scale, int long_from, void* stream);

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
// second form (csrc/gemv_mfma2.hip): activations in registers, two workgroups per CU, coalesced 4-bit loads; the RMSNorm
// prologue runs as one rows kernel into the workspace scratch at VLM_MFMA_WS_XN_OFFSET.  -1: shape not handled.
#define VLM_MFMA_WS_XN_OFFSET ((size_t)4096 * 256 * 4 + 8192 * 4)
#define VLM_MFMA_WS_XN_BYTES ((size_t)16 * 8192 * 2)
VLM_INTERNAL int vlm_gemv_mfma2_try_bf16(const void* x, const void* W, const void* Wsb, const void* bias, const void* res,
                                         const void* norm_w, void* y, int M, int N, int K, int ldx, int ldw, int ldy, int ldres,
                                         float eps, int epilogue, const VlmRopeKv* rk, void* ws, void* stream);
VLM_INTERNAL int vlm_gemv_mfma2_try_w4(const void* x, const void* W, const void* Wsb, const void* bias, const void* res,
                                       const void* norm_w, void* y, int M, int N, int K, int ldx, int ldw, int ldy, int ldres,
                                       float eps, int epilogue, const VlmRopeKv* rk, void* ws, void* stream);
// long-K form (csrc/gemv_mfma_longk.hip): bf16, no norm prologue, few row tiles; -1: shape not handled
VLM_INTERNAL int vlm_gemv_mfma_longk_try(const void* x, const void* W, const void* bias, const void* res, void* y, int M, int N,
                                         int K, int ldx, int ldw, int ldy, int ldres, int epilogue, void* stream);
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

// row-slice form of the skinny-M decode GEMM (csrc/gemv_mfma_rows.hip): x in the tiled layout [K / 8][16][8]; -1: shape not handled
VLM_INTERNAL int vlm_gemv_mfma_rows_ok(int M, int N, int K);      // 1: vlm_gemv_mfma_rows_try takes this projection
VLM_INTERNAL int vlm_gemv_mfma_rows_try(const void* xt, const void* W, const void* bias, const void* res, void* y, int M, int N, int K,
                                        int ldw, int ldy, int ldres, int epilogue, void* stream);

VLM_INTERNAL int vlm_sample_last_launches(void);   /* sample.hip: kernels the last vlm_sample / vlm_sample_advance of this thread enqueued */

/* sample.hip: the sampled step's tail for the engine's captured step - vlm_sample (temperature > 0; top_p / min_p / top_k) with the
 * final pick, vlm_decode_advance and the next step's embedding gather in ONE last launch */
VLM_INTERNAL int vlm_sample_advance(const void* logits, int ld, int B, int V, void* logprobs, void* scratch, int ldlp, void* tok,
                                    void* workspace, float temperature, float top_p, float min_p, int top_k, unsigned seed,
                                    void* ctx, void* pos, void* out_ring, int ring_len, void* step, const void* embed, void* h,
                                    int D, int ldh, void* stream);


/* gemm_bf16.hip: what the reduce launch of a split-K GEMM does ON TOP of its epilogue (the wide decode steps of engine.hip): the
 * launch the caller would otherwise issue next on the reduced rows.  Bit-identical to the two-launch sequence. */
enum { VLM_TAIL_NONE = 0, VLM_TAIL_RMSNORM = 1, VLM_TAIL_ROPE_KV = 2 };
struct VlmGemmTail {
  int kind;
  /* VLM_TAIL_RMSNORM: xn[m] = norm_w * T(C[m] * rsqrt(mean(C[m]^2) + eps)) (vlm_rmsnorm_residual on the rows just written) */
  const void* norm_w;
  float eps;
  void* xn;
  int ldxn;
  /* VLM_TAIL_ROPE_KV: vlm_mrope_kvwrite_decode on the qkv rows just written (row b at text position pos[b], slot[b] of
   * block-table row b) */
  int Hq, Hkv, D;
  const int* pos;
  const float* inv_freq;
  const int* slot;
  const int* block_table;
  int max_pages;
  unsigned short* kpool;
  unsigned short* vpool;
  float qk_scale;
  int long_from;
};
VLM_INTERNAL int vlm_gemm_bf16_tail(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K,
                                    int lda, int ldw, int ldc, int ldres, int epilogue, const VlmGemmTail* tail, int* tail_done,
                                    void* stream);
VLM_INTERNAL int vlm_gemm_w4_tail(const void* A, const void* Wq, const void* Wsb, const void* bias, const void* res, void* C, int M,
                                  int N, int K, int lda, int ldc, int ldres, int epilogue, const VlmGemmTail* tail, int* tail_done,
                                  void* stream);
