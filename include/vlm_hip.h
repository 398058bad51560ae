/* vlm_hip.h - C ABI of libvlm_hip.so, the MI355X (gfx950) operator library that
 * replaces the MLX ops on mlx-vlm's generate hot path (Qwen2-VL first).
 *
 * The reference (Blaizzy/mlx-vlm v0.6.15, /root/reference) has NO FFI/plugin ABI
 * of its own: its model code calls straight into the `mlx` Python runtime.  The
 * drop-in boundary is therefore the set of MLX operators that path invokes
 * (SURVEY.md par. 8b); every entry point below names the reference call site it
 * replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless marked (host); the caller owns
 *     every buffer; bf16 tensors are raw 16-bit words; "ld*" / "*_stride" are
 *     leading dimensions in ELEMENTS
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     every function only enqueues work and returns
 *   - return value: 0 = VLM_OK; 1 = bad argument; 2 = unsupported shape;
 *     1000 + hipError_t = a HIP launch error.  No exceptions cross the ABI.
 *   - no global state; functions are re-entrant; one stream per engine thread
 *     (the reference's contract: one thread-local generation stream,
 *     mlx_vlm/generate/common.py:32)
 */
#ifndef VLM_HIP_H_
#define VLM_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* epilogue flags for vlm_gemm_bf16 / vlm_gemv_bf16 (rounded to bf16 at the same
 * points as the reference's typed graph: after bias, after activation, after
 * the residual add) */
#define VLM_EPI_NONE 0
#define VLM_EPI_BIAS 1       /* + bias[n]                           nn.Linear(bias=True) */
#define VLM_EPI_GELU_FAST 2  /* x*sigmoid(1.702x)                   vision.py:167        */
#define VLM_EPI_GELU_ERF 4   /* exact erf GELU                      vision.py:112        */
#define VLM_EPI_RESIDUAL 8   /* + res[m][n]                         vision.py:188-193, language.py:151-153 */
#define VLM_EPI_SWIGLU 16    /* W rows interleaved (gate_j, up_j): out[j] = silu(g)*u   mlp.py:6-14, activations.py:7-9 */
/* vlm_gemv_bf16_ws only, batched decode steps of 5..16 rows (ABI v8): the activations BETWEEN two projections in the tiled layout
 * t[K / 8][16][8] bf16 (k group, batch row, 8 consecutive k) instead of row-major [M][K] - what the matrix cores' B fragments read
 * as whole cache lines.  Y_TILED: y is written tiled (ldy ignored; y holds 16 * N_out elements, rows M..15 are not written);
 * X_TILED: x is read tiled (ldx ignored) - the row-slice form of the skinny decode GEMM (csrc/gemv_mfma_rows.hip), for projections
 * with at most 16 output rows per compute unit and K >= 4096, K % 128 == 0 (the SwiGLU MLP's down projection, mlp.py:6-14).
 * A shape the flagged kernel does not take is refused (return 2) and nothing is enqueued. */
#define VLM_EPI_X_TILED 64
#define VLM_EPI_Y_TILED 128

int vlm_abi_version(void);

/* C[M,N] = epi(A[M,K] . W[N,K]^T), bf16 in / fp32 MFMA accumulate / bf16 out.
 * Replaces nn.Linear (mlx_vlm/models/qwen2_vl/vision.py:110-120,129-130,137,161,168-173;
 * language.py:52-55,76,120; models/mlp.py:9-14), embed_tokens.as_linear / lm_head
 * (language.py:514-517) and the Conv3d patch projection (vision.py:83-101).
 * K % 8 == 0, N % 8 == 0.  With VLM_EPI_SWIGLU, C is [M, N/2]. */
int vlm_gemm_bf16(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K,
                  int lda, int ldw, int ldc, int ldres, int epilogue, void* stream);

/* The qkv projection of the vision attention with apply_rotary_pos_emb_vision (vision.py:35-50,141-142) as its epilogue:
 * C = rope2d(A . W^T + bias) on the first rope_cols columns (the q and k heads, head_dim wide), plain bias on the rest.
 * W's q / k rows must be INTERLEAVED per head - row (h, d) of the checkpoint at h * head_dim + 2 d, row (h, d + hd/2) at
 * h * head_dim + 2 d + 1 (q.k is invariant under a common permutation of the head dimension) - so that a rotation pair
 * is two adjacent outputs.  cos_sin: fp32 [2][M][head_dim / 2]: cos rows then sin rows of VisionModel.rot_pos_emb
 * (vision.py:219-255).  Rounding as the reference: bf16(linear + bias), fp32 rotation, one rounding. */
int vlm_gemm_bf16_rope2d(const void* A, const void* W, const void* bias, const void* cos_sin, void* C, int M, int N, int K,
                         int lda, int ldw, int ldc, int head_dim, int rope_cols, void* stream);

/* The same GEMM over MLX affine 4-bit weights, dequantisation FUSED into the W-tile staging (nn.QuantizedLinear /
 * mx.quantized_matmul at L > 1, reference utils.py:918-967): Wq uint32 [N][K/8] (element k of a row in word k / 8, nibble
 * k % 8), Wsb uint32 [N][K/64] (scale bf16 | bias bf16 << 16 per 64-wide group), K % 64 == 0.  The LDS image of a W tile
 * is the one vlm_gemm_bf16 builds from vlm_dequant_w4's output, so C is bit-identical to dequantise-then-GEMM while the
 * weights move at 4.5 bits instead of being written and re-read at 16.  Epilogues: NONE, BIAS, RESIDUAL, BIAS | RESIDUAL,
 * SWIGLU. */
int vlm_gemm_w4(const void* A, const void* Wq, const void* Wsb, const void* bias, const void* res, void* C, int M, int N,
                int K, int lda, int ldc, int ldres, int epilogue, void* stream);

/* test / A-B knob for vlm_gemm_bf16 kernel selection: 0 = automatic (LDS-DMA staging when K % 64 == 0; the phased
 * 256x256 kernel from ~120 tiles up), 1 = 128x128 kernel with global -> VGPR -> LDS staging, 2 = 128x128 kernel with
 * LDS-DMA staging, 3 = 256x256 phased kernel whenever legal (K % 64 == 0, K >= 128, no SwiGLU), 4 = its 2-phase
 * variant, 5 = automatic with the 2-phase variant, 6 / 7 = as 3 with 256x192 / 256x256 tiles forced, 8 = as 2 (named
 * "no split-K"), 9 = 128 kernel family with split-K x4 forced, 10 = as 3 with the persistent tile loop.  Modes 1-8 are bit-identical to each other; split-K (automatic
 * for few tiles and K >= 2048, or mode 9) sums fp32 partials of K ranges and agrees to fp32 summation order.
 * 100 + 10 * splits + cfg (round 6, tile sweeps - scripts/r06/gemm_tiles.py): the plain kernels only, tile cfg 1 = 64 x 64,
 * 2 = 64 x 128, 3 = 128 x 128 (0 = the policy), split-K forced to `splits` K ranges (0 / 1 = never).
 * The automatic policy since round 6: up to 64 rows (wide decode steps, short prompts) 64 x 64 tiles - 32 x 64 up to 32 rows -
 * and split-K from K = 1024, because such a launch is a weight stream and wants workgroups (VLM_GEMM_SKINNY64=0 /
 * VLM_GEMM_SKINNY32=0 in the environment restore the policy of rounds 1-5 for A/B runs). */
int vlm_gemm_set_staging(int mode);

/* y[M,N] = epi(x[M,K] . W[N,K]^T) for the decode step, M in {1,2,4,8}; weight streaming.
 * norm_w != NULL fuses y = f(RMSNorm(x; norm_w, eps)) (language.py:130-133,149-153,200).
 * Same reference call sites as vlm_gemm_bf16 at L == 1. */
int vlm_gemv_bf16(const void* x, const void* W, const void* bias, const void* res, const void* norm_w, void* y, int M,
                  int N, int K, int ldx, int ldw, int ldy, int ldres, float eps, int epilogue, void* stream);

/* Decode-step fusions built on the same kernels (one decoder layer = 5 launches):
 * vlm_gemv_qkv_rope_kvwrite: qkv = RMSNorm(h) Wqkv^T + b (language.py:52-54,76,149), then M-RoPE on the q and k
 *   heads at pos[m] (rope_utils.py:567-651; a decoded text token has equal t/h/w positions, language.py:476-509)
 *   and KVCache.update_and_fetch (cache.py:345-367): rotated q -> qkv[m][0 : Hq*D], rotated k and v -> slot[m] of
 *   sequence m in the paged pools.  pos / slot int32 [M] on the device.  block_table == NULL selects the identity
 *   layout: row m owns pages [m * max_pages, (m + 1) * max_pages) of the pools as passed (no table load).
 * vlm_gemv_attn_out: h += merge(attention split partials) Wo^T (language.py:115-120,151): the split-K merge of
 *   vlm_attn_decode_paged is the GEMV prologue, the residual add its epilogue. */
int vlm_gemv_qkv_rope_kvwrite(const void* h, const void* norm_w, float eps, const void* Wqkv, const void* bqkv,
                              void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D, const void* pos,
                              const void* slot, const void* inv_freq, const void* block_table, int max_pages,
                              void* kpool, void* vpool, void* stream);

/* The same projections for a batched decode step of up to 16 rows: from 5 rows on (9 for the qkv form) the batch rows
 * become the N dimension of v_mfma_f32_16x16x32_bf16 (csrc/gemv_mfma.hip) - one weight stream for all rows instead of M dot
 * products per weight chunk.  `workspace` (vlm_gemv_workspace_bytes() bytes, ZERO-INITIALISED once, owned by the caller;
 * launches sharing it must be ordered by one stream) lets projections with few output rows and a long K (o_proj, down)
 * split K over workgroups: fp32 partial tiles + an arrival ticket per tile, summed in a fixed order (deterministic).
 * NULL = no split; shapes that then do not fit fall back to the kernels of vlm_gemv_bf16 (M in {1, 2, 4, 8}). */
size_t vlm_gemv_workspace_bytes(void);
int vlm_gemv_bf16_ws(const void* x, const void* W, const void* bias, const void* res, const void* norm_w, void* y, int M, int N,
                     int K, int ldx, int ldw, int ldy, int ldres, float eps, int epilogue, void* workspace, void* stream);
int vlm_gemv_qkv_rope_kvwrite_ws(const void* h, const void* norm_w, float eps, const void* Wqkv, const void* bqkv, void* qkv, int ldq,
                                 int M, int hidden, int Hq, int Hkv, int D, const void* pos, const void* slot,
                                 const void* inv_freq, const void* block_table, int max_pages, void* kpool, void* vpool,
                                 void* workspace, void* stream);
int vlm_gemv_attn_out(const void* part_o, const void* part_ml, int nsplit, const void* Wo, void* h, int ldh, int M,
                      int N, int Hq, int D, void* stream);

/* MLX affine 4-bit weights (group 64): what load_model turns every Linear / Embedding of a 4-bit checkpoint into
 * (utils.py:918-967: nn.quantize with the `<path>.scales in weights` predicate -> nn.QuantizedLinear = mx.quantized_matmul(x,
 * weight, scales, biases, transpose=True, group_size=64, bits=4) (+ bias); nn.QuantizedEmbedding = mx.dequantize of the rows).
 * Engine layout (repacked once at load): Wq uint32 [N][K/8], MLX's own words (element k of a row in word k / 8, bits
 * 4 (k % 8) .. +3); Wsb uint32 [N][K/64] = scale bf16 (low half) | bias bf16 (high half) of each group.  K % 64 == 0.
 * vlm_gemv_w4 / vlm_gemv_w4_qkv_rope_kvwrite: the decode GEMVs of vlm_gemv_bf16 / vlm_gemv_qkv_rope_kvwrite with the
 *   dequantisation fused into the operand path (fp32 affine form per group, no weight is rounded); M in {1,2,4,8}.
 * vlm_dequant_w4: rows of the matrix as bf16 (fp32 scale * q + bias, one rounding): rows == NULL -> rows 0..n_rows-1
 *   (prefill: materialised once per projection, then vlm_gemm_bf16), rows != NULL -> gather (QuantizedEmbedding lookup). */
int vlm_gemv_w4(const void* x, const void* Wq, const void* Wsb, const void* bias, const void* res, const void* norm_w, void* y,
                int M, int N, int K, int ldx, int ldy, int ldres, float eps, int epilogue, void* stream);
int vlm_gemv_w4_qkv_rope_kvwrite(const void* h, const void* norm_w, float eps, const void* Wq, const void* Wsb, const void* bqkv,
                                 void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D, const void* pos, const void* slot,
                                 const void* inv_freq, const void* block_table, int max_pages, void* kpool, void* vpool,
                                 void* stream);
/* batched decode rows (5..16; 9..16 for the qkv form) over 4-bit weights: the dequant-fused MFMA form of csrc/gemv_mfma.hip -
 * nibbles become bf16 128 + q in the A fragments, every 64-wide group enters the sum as scale * (D - 128 sum_x) + bias * sum_x in
 * fp32 (no weight is rounded); `workspace` as vlm_gemv_bf16_ws.  Other row counts: the kernels of vlm_gemv_w4. */
int vlm_gemv_w4_ws(const void* x, const void* Wq, const void* Wsb, const void* bias, const void* res, const void* norm_w, void* y,
                   int M, int N, int K, int ldx, int ldy, int ldres, float eps, int epilogue, void* workspace, void* stream);
int vlm_gemv_w4_qkv_rope_kvwrite_ws(const void* h, const void* norm_w, float eps, const void* Wq, const void* Wsb, const void* bqkv,
                                    void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D, const void* pos, const void* slot,
                                    const void* inv_freq, const void* block_table, int max_pages, void* kpool, void* vpool,
                                    void* workspace, void* stream);
int vlm_dequant_w4(const void* Wq, const void* Wsb, const void* rows, void* out, int n_rows, int K, int ldo, int n_table_rows,
                   void* stream);

/* nn.LayerNorm(eps) -> mx.fast.layer_norm (vision.py:109,180-181). dim % 8 == 0, dim <= 8192 */
int vlm_layernorm(const void* x, const void* w, const void* b, void* y, int rows, int dim, float eps, void* stream);

/* nn.RMSNorm -> mx.fast.rms_norm (language.py:130-133,168) with the residual add of
 * language.py:151-153 fused: h = x + res (res may be NULL), y = rmsnorm(h) * w, h_out = h (may be NULL) */
int vlm_rmsnorm_residual(const void* x, const void* res, const void* w, void* y, void* h_out, int rows, int dim,
                         float eps, void* stream);

/* apply_rotary_pos_emb_vision (vision.py:35-50) in place on q and k of qkv [N][3][H][D];
 * cos_tab / sin_tab fp32 [N][D/2] (cos / sin of VisionModel.rot_pos_emb, vision.py:219-255) */
int vlm_rope2d_vision(void* qkv, const void* cos_tab, const void* sin_tab, int N, int H, int D, int ld, void* stream);

/* M-RoPE (MRoPERotaryEmbedding.apply_rotary, fused-kernel numerics: rope_utils.py:567-651,
 * 1243-1286, selector 519-526) applied in place to the q and k heads of
 * qkv [T][(Hq + 2 Hkv) * D], fused with KVCache.update_and_fetch (cache.py:345-367): the rotated
 * k and the v of token t are written to slot kv_slot[t] of sequence kv_seq[t] (NULL: seq = t) in
 * the paged pools (page = block_table[seq][slot / 64]).  pos_* int32 [T] (t, h, w axes; pass the
 * same pointer three times for text).  sec0 / sec1 = mrope_section[0..1].  kpool == NULL: rope only.
 * K pool [page][Hkv][D/8][64][8], V pool [page][Hkv][D][64 key slots] (bf16; slot order: see csrc/common.hpp
 * vlm_vslot - the k-slot order of the decode P.V MFMA). */
int vlm_mrope_kvwrite(void* qkv, int ld, int T, int Hq, int Hkv, int D, const void* pos_t, const void* pos_h,
                      const void* pos_w, const void* inv_freq, int sec0, int sec1, const void* kv_seq,
                      const void* kv_slot, const void* block_table, int max_pages, void* kpool, void* vpool,
                      void* stream);
/* the same with SuScaledRoPE's input scale (rope_utils.py:174-176, Phi-3.5 phi3_v.py:41-48): q and k are multiplied by
 * qk_scale and rounded to bf16 (the reference's typed `scale.astype(x.dtype) * x`) before the rotation; v is not scaled.
 * qk_scale = 1 is vlm_mrope_kvwrite bit for bit. */
int vlm_mrope_kvwrite_scaled(void* qkv, int ld, int T, int Hq, int Hkv, int D, const void* pos_t, const void* pos_h,
                             const void* pos_w, const void* inv_freq, int sec0, int sec1, const void* kv_seq,
                             const void* kv_slot, const void* block_table, int max_pages, void* kpool, void* vpool,
                             float qk_scale, void* stream);

/* The fetch half of KVCache.update_and_fetch (cache.py:345-367 returns keys / values [..., :offset, :]) over the PAGED
 * pools: the cached (rotated) k and v of token t - slot kv_slot[t] of sequence kv_seq[t] (NULL: seq = t) - are copied
 * into the k / v columns of row t of a token-major qkv buffer [T][(Hq + 2 Hkv) * D] (the q columns are left alone).
 * Used when a prompt chunk is prefilled onto a non-empty cache (ar.py:426-472 chunked prefill, dispatch.py:861-882
 * prompt_cache continuation): the chunk attends to [cached tokens | its own]. */
int vlm_kv_gather(void* qkv, int ld, int T, int Hq, int Hkv, int D, const void* kv_seq, const void* kv_slot,
                  const void* block_table, int max_pages, const void* kpool, const void* vpool, void* stream);

/* mx.fast.scaled_dot_product_attention on the prefill path: varlen segments (cu_seqlens int32
 * [nseg+1]), mask=None (vision.py:148-158) or mask="causal" (base.py:214-228,366-373), GQA.
 * q/k/v/out are token-major with the given token strides; head h of token t at ptr + t*stride + h*D.
 * total_qblocks = sum_s ceil(len_s / 128) (host).  D in {64, 80, 128}.
 * causal: bit 0 = causal mask; bit 1 = hint "all segments have the same length" (enables an XCD-local
 * placement of the query blocks of one (segment, head); results are identical with or without it);
 * bit 2 = cu_seqlens is followed by q_start int32 [nseg]: rows of segment s before q_start[s] are KEYS ONLY (the
 * cached prefix of a prompt chunk appended to a non-empty cache, ar.py:426-472 / dispatch.py:861-882 - the causal
 * mask stays on absolute rows, i.e. offset by the cache length as base.py:366-373 builds it); their rows of `out`
 * are not written and total_qblocks = sum_s ceil((len_s - q_start_s) / 128).  Not with bit 1. */
int vlm_attn_prefill(const void* q, const void* k, const void* v, void* out, int q_stride, int k_stride, int v_stride,
                     int o_stride, const void* cu_seqlens, int nseg, int total_qblocks, int Hq, int Hkv, int D,
                     float scale, int causal, void* stream);

/* the same op at L == 1 over the paged cache (base.py:366-373 from language.py:115-118).
 * q [B][Hq*D] (row stride ldq); kv_len int32 [B] (+ kv_len_add) keys per sequence;
 * part_o fp32 [B][Hq][nsplit][D], part_ml fp32 [B][Hq][nsplit][2] receive the per-split partials; out [B][Hq*D]
 * (bf16) gets the merged result, or pass out == NULL and merge in vlm_gemv_attn_out.  D == 128.
 * block_table == NULL: identity layout - sequence b owns pages [b * max_pages, (b + 1) * max_pages) of the pools as
 * passed, page numbers are computed instead of loaded (one dependent memory round trip less). */
int vlm_attn_decode_paged(const void* q, int ldq, const void* kpool, const void* vpool, const void* block_table,
                          int max_pages, const void* kv_len, int kv_len_add, int B, int Hq, int Hkv, int D,
                          float scale, int nsplit, void* part_o, void* part_ml, void* out, int ldo, void* stream);

/* The same op (base.py:366-373 at L == 1) in its page-split form, the one the decode engine uses for up to 64
 * (sequence, kv head) pairs: nsplit one-wave workgroups per (sequence, kv head), workgroup s walks pages s, s + nsplit,
 * ... - one 64-token page (32 KB of K + V) is the whole load of a compute unit, so a context is pulled through
 * min(pages, nsplit) CUs per kv head instead of one - and the LAST workgroup to arrive (one agent-scope ticket per
 * (sequence, kv head)) merges the partials and writes out [B][Hq*D] (bf16): one launch, no merge pass, no merge in the
 * o_proj prologue.  part_o fp32 [B*Hkv][nsplit][Hq/Hkv][D] and part_ml fp32 [B*Hkv][nsplit][Hq/Hkv][2] are scratch
 * (same sizes as vlm_attn_decode_paged's); tickets: uint32 [B*Hkv], ZERO before the first launch - every launch leaves
 * it zero again.  Launches that share part_o / part_ml / tickets must be stream-ordered. */
/* out == NULL: the partial-only form.  Nothing is merged and no ticket is taken (tickets may be NULL): every split s
 * of every head writes (m, l) - m in the log2 domain, (-inf, 0) for a split with no page - to part_ml fp32
 * [B][Hq][nsplit][2] and, when it has pages, its unnormalised O as fp32 to part_o [B][Hq][nsplit][D] (ABI v8; bf16 before -
 * a rounding point the reference's fused attention does not have); the launch ends at those (plain) stores and
 * vlm_gemv_attn_out_bf16 merges them in the o_proj prologue. */
int vlm_attn_decode_paged_split(const void* q, int ldq, const void* kpool, const void* vpool, const void* block_table,
                                int max_pages, const void* kv_len, int kv_len_add, int B, int Hq, int Hkv, int D,
                                float scale, int nsplit, void* part_o, void* part_ml, void* tickets, void* out, int ldo,
                                void* stream);

/* Uniform 8-bit KV cache: QuantizedKVCache (cache.py:233-334, mx.quantize bits = 8, group_size = 64 over the head
 * dimension), KVCache.to_quantized (cache.py:415-423) and quantized_scaled_dot_product_attention at L == 1
 * (base.py:260-302).  8-bit pools per layer, same pages / block table as the bf16 pools:
 *   kpool8 u8 [page][Hkv][D/8][64 keys][8], vpool8 u8 [page][Hkv][D][64 key slots] (the bf16 layouts with 1-byte elements),
 *   ksb / vsb uint32 [page][Hkv][64 keys][D/64]: (scale bf16 | bias bf16 << 16) of key k, group j at word k * (D/64) + j.
 * layer_stride: ELEMENTS between consecutive layers' pools, identical for the bf16 and the u8 pools (the sb pools advance
 * by layer_stride / (D/2) words).  D == 128.
 * vlm_kv_quantize_tokens: token i of the list (slot kv_slot[i] of sequence kv_seq[i]; kv_seq NULL: row i) is quantised from
 *   the bf16 pools into the 8-bit pools, all n_layers layers - to_quantized over a range of cached tokens.
 * vlm_attn_decode_paged_q8: the page-split decode attention (vlm_attn_decode_paged_split: same nsplit / part_o / part_ml /
 *   tickets / out conventions, out == NULL = partial-only form) over the 8-bit pools; quantize_new != 0: the step's new
 *   token (slot kv_len - 1, already written to the bf16 pools by the qkv epilogue) is quantised first -
 *   QuantizedKVCache.update_and_fetch - by the workgroup that owns its page.  In the merging form (out != NULL) a step of
 *   128 or more (row, kv head) pairs with nsplit <= 16 runs half-page units: part_o / part_ml must then hold 2 * nsplit
 *   splits per (row, head). */
int vlm_kv_quantize_tokens(const void* kpool, const void* vpool, void* kpool8, void* vpool8, void* ksb, void* vsb,
                           size_t layer_stride, int n_layers, const void* kv_seq, const void* kv_slot, int T,
                           const void* block_table, int max_pages, int Hkv, int D, void* stream);
int vlm_attn_decode_paged_q8(const void* q, int ldq, const void* kpool16, const void* vpool16, void* kpool8, void* vpool8,
                             void* ksb, void* vsb, const void* block_table, int max_pages, const void* kv_len,
                             int kv_len_add, int B, int Hq, int Hkv, int D, float scale, int nsplit, void* part_o,
                             void* part_ml, void* tickets, void* out, int ldo, int quantize_new, void* stream);

/* h[0][0:N] += merge(partials of vlm_attn_decode_paged_split's partial-only form) Wo^T for ONE decode row
 * (language.py:115-120,151): every thread of the o_proj GEMV loads one 8-element chunk of all nsplit <= 16 fp32 partials
 * and their (m, l) ahead of its weight stream and merges them in registers (bf16 Wo: the name).  Hq * D <= 2048. */
int vlm_gemv_attn_out_bf16(const void* part_o, const void* part_ml, int nsplit, const void* Wo, void* h, int ldh, int N,
                           int Hq, int D, void* stream);

/* nn.Embedding (language.py:164,179): out[t] = table[ids[t]] */
int vlm_embed_gather(const void* ids, const void* table, void* out, int T, int D, int ldo, int vocab, void* stream);

/* Model.merge_input_ids_with_image_features (qwen2_vl.py:78-148): dst[dst_rows[i]] = src[i] */
int vlm_scatter_image_rows(const void* src, const void* dst_rows, void* dst, int n, int D, int ld_src, int ld_dst,
                           void* stream);

/* pixel_values.astype(bf16) (qwen2_vl.py:44-45) with zero padding of the row to ld_dst */
int vlm_cast_f32_bf16_pad(const void* src, void* dst, int rows, int cols, int ld_src, int ld_dst, void* stream);

/* decode-loop bookkeeping kept on the device (cache.py:362 offset += 1; language.py:476-509
 * pos = offset + rope_delta): ctx[b]++, pos[b]++, out_ring[step % ring_len][b] = tok[b], step++ */
int vlm_decode_advance(void* ctx, void* pos, const void* tok, void* out_ring, int ring_len, void* step, int B,
                       void* stream);

/* ar.py:368 logprobs = logits - logsumexp (bf16) -> logprobs [B][V] (may be NULL when greedy);
 * temperature == 0: argmax, lowest index on ties (sample_utils.py:63-64);
 * else top_p (289-318) -> min_p (266-286) -> top_k (169-175) -> categorical(logprobs/temp) (385-387)
 * by Gumbel-max with a counter-hash RNG keyed by (seed, *step_ptr, row, index).
 * scratch bf16 [B][ldlp] is required when temperature > 0.
 * workspace: vlm_sample_workspace_bytes(B) bytes, zero-filled ONCE at allocation and then owned by the sampler: an arrival
 * ticket, the per-block argmax partials, a 65 536-bin key histogram per row (every call leaves it all-zero again) and 128
 * control words per row for the top-p path that is split over several workgroups (crossing key, per-slice counts, the
 * populated key range - re-armed by the last launch of a call; the first call after the zero fill sees range [0, kmax],
 * takes the unwindowed route and gives the same row).  One workspace serves one stream at a time. */
size_t vlm_sample_workspace_bytes(int B);
int vlm_sample(const void* logits, int ld, int B, int V, void* logprobs, void* scratch, int ldlp, void* tok,
               void* workspace, float temperature, float top_p, float min_p, int top_k, unsigned seed,
               const void* step_ptr, void* stream);

/* The whole sampler surface of make_sampler (sample_utils.py:10-89), filters in its order: top_n_sigma (181-212) -> p_less
 * (215-236) -> typical_p (321-345) -> top_p (289-318) -> min_p with min_tokens_to_keep (239-286) -> xtc (348-376) -> top_k
 * (169-175) -> categorical(logprobs / temp) (385-387).  The python scalars arrive as doubles and are converted the way MLX's
 * weak typing converts them (double -> float32 -> the log-probs' dtype) before they meet the bf16 log-probs; every
 * elementary op of the typed graph rounds to bf16 (pinned by tests/golden/samplers_ref.npz).  A filter is off at the value
 * make_sampler treats as off: top_p outside (0, 1), min_p == 0, top_k <= 0, top_n_sigma == 0, p_less == 0, typical_p outside
 * (0, 1), xtc_probability == 0.  After the call `scratch` holds the filtered log-probs (-inf = removed) when at least one filter
 * is on; with none the draw reads the log-probs themselves.  Launches: logsumexp partials, log-probs, [filters: one workgroup
 * per row], the Gumbel-max partials (64 workgroups per row), the final pick.
 * typical_p needs sort_workspace (device, vlm_sample_sort_workspace_bytes(B, V)).  xtc: B == 1 only (the reference's minimum
 * runs over the whole array and its draw is one scalar per call) -> VLM_ERR_SHAPE otherwise; the draw is the counter hash at
 * (seed, *step_ptr, row, 0xFFFFFFFF); xtc_special_tokens = device int32 [n_xtc_special <= 256]. */
typedef struct vlm_sampler_params {
  double temperature, top_p, min_p;
  int min_tokens_to_keep, top_k;
  double top_n_sigma;
  int p_less;
  double typical_p, xtc_probability, xtc_threshold;
  const void* xtc_special_tokens;
  int n_xtc_special;
  void* sort_workspace;
  unsigned seed;
  int input_is_logprobs; /* temperature > 0: `logits` already holds log-probs (what the reference hands a sampler closure,
                            ar.py:368-379): no logsumexp pass, `logprobs` may be NULL */
} vlm_sampler_params;
size_t vlm_sample_sort_workspace_bytes(int B, int V);
int vlm_sample_ex(const void* logits, int ld, int B, int V, void* logprobs, void* scratch, int ldlp, void* tok, void* workspace,
                  const vlm_sampler_params* p, const void* step_ptr, void* stream);

/* Logits processors of generate_step (make_logits_processors, sample_utils.py:92-146, applied at ar.py:360-364) as ONE
 * device pass over the logits row(s), in the reference's order: logit_bias (129-134) -> repetition penalty (390-422:
 * x < 0 ? x * p : x / p, once per distinct token of the last rep_ctx fed tokens) -> presence penalty (425-450: - p once
 * per distinct token) -> frequency penalty (453-475: - p per occurrence).  Every step rounds to bf16 as the reference's
 * typed graph does (python-float penalties are rounded to bf16 first).
 * The history of fed tokens (prompt + every token fed back) lives on the device so the step stays graph-replayable:
 * hist int32 [B][hist_cap] ring, hist_len int32 [B] tokens pushed so far.  push_tok != NULL: push_tok[b] is appended to
 * row b's history FIRST (the decode step feeds tok[b]; the reference appends y before the processors run).
 * A penalty of 0 / a context of 0 switches that processor off.  Contexts <= hist_cap <= 1024. */
typedef struct vlm_penalty_args {
  void* hist;
  void* hist_len;
  int hist_cap;
  float rep_penalty;
  int rep_ctx;
  float pres_penalty;
  int pres_ctx;
  float freq_penalty;
  int freq_ctx;
  const void* bias_idx; /* int32 [n_bias] (distinct) */
  const void* bias_val; /* fp32 [n_bias] */
  int n_bias;
  /* per-request processors of a continuous batch (BatchGenerator.insert(..., logits_processors=), ar.py:2584-2606):
   * row_params != NULL: fp32 [B][8] on the device = {rep_penalty, rep_ctx, pres_penalty, pres_ctx, freq_penalty, freq_ctx,
   * n_bias, 0} of row b - the scalar fields above are then ignored - and row b's bias list sits at bias_idx / bias_val +
   * b * bias_stride.  The host rewrites the table as requests join and leave; a captured step keeps its arguments. */
  const void* row_params;
  int bias_stride;
} vlm_penalty_args;
int vlm_apply_logit_penalties(void* logits, int ld, int B, int V, const void* push_tok, const vlm_penalty_args* p,
                              void* stream);

/* The greedy tail of a decode step in two launches instead of five: vlm_sample at temperature 0 (ar.py:368 logprobs,
 * sample_utils.py:63-64 argmax) + vlm_decode_advance (cache.py:362, language.py:476-509) + the NEXT step's
 * vlm_embed_gather (language.py:164,179): h[b] = embed[tok[b]] (row stride ldh).  workspace: vlm_sample_workspace_bytes,
 * zero-filled once at allocation (its first word is an arrival ticket the kernel re-arms). */
int vlm_sample_greedy_advance(const void* logits, int ld, int B, int V, void* logprobs, int ldlp, void* tok,
                              void* workspace, void* ctx, void* pos, void* out_ring, int ring_len, void* step,
                              const void* embed, void* h, int D, int ldh, void* stream);

/* ------------------------------------------------------------------------------------------
 * Model-level engine: the layer loops of Qwen2Model / VisionModel run natively so that a decode
 * step is ~150 back-to-back launches with no host code in between and can be captured in a
 * hipGraph (the reference keeps the loop in Python and relies on MLX's lazy graph + async_eval,
 * mlx_vlm/generate/ar.py:474-508).
 * ------------------------------------------------------------------------------------------ */
typedef struct vlm_llm_config {
  int hidden, n_layers, inter, n_heads, n_kv_heads, head_dim, vocab;
  float rms_eps;
  int mrope_sec0, mrope_sec1; /* mrope_section[0], [1] */
  float attn_scale;           /* softmax scale of the attention; 0 = head_dim ** -0.5.  Set when the engine's head_dim is a
                                 zero-padded form of the model's (e.g. 64 -> 128, llava_bunny language.py:24-25) */
  float rope_qk_scale;        /* ABI v3.  SuScaledRoPE (rope_utils.py:96-189): q and k are multiplied by this and rounded to
                                 bf16 before the rotation, in the prefill pass and in every decode qkv epilogue; 0 = 1 = none */
  int rope_long_from;         /* ABI v4.  SuScaledRoPE's short / long factor regimes: > 0 = original_max_position_embeddings and
                                 vlm_llm_globals.inv_freq holds TWO tables [2][head_dim / 2] (short factors, then long).  A decode
                                 step picks the long table for every row when ANY row's cache offset >= this value - the
                                 reference's per-call rule position_end = max(offset) + 1 > original_max (rope_utils.py:168-172)
                                 evaluated inside the qkv epilogue, so captured steps cross the limit without host help and a
                                 continuous batch behaves as the reference's batched call; a prefill call names its regime in
                                 vlm_prefill_args.rope_long.  0 = one table */
} vlm_llm_config;

typedef struct vlm_llm_layer {
  const void *ln1_w, *wqkv, *bqkv, *wo, *ln2_w, *wgu /* interleaved gate/up rows */, *wdown;
  /* MLX affine 4-bit layers (vlm_gemv_w4): a non-NULL *_sb marks the matching w* as q words uint32 [N][K/8] with *_sb its
   * (scale | bias << 16) words uint32 [N][K/64]; NULL = bf16 weight as before.  Rows packed / interleaved the same way. */
  const void *wqkv_sb, *wo_sb, *wgu_sb, *wdown_sb;
} vlm_llm_layer;

typedef struct vlm_llm_globals {
  const void *embed, *final_norm_w, *lm_head /* == embed when tied */, *inv_freq /* fp32 [head_dim/2] */;
  const void *embed_sb, *lm_head_sb; /* 4-bit embedding table / head (nn.QuantizedEmbedding, as_linear): see vlm_llm_layer */
} vlm_llm_globals;

typedef struct vlm_kv_pool {
  void *kpool, *vpool;      /* layer 0 base */
  size_t layer_stride;      /* elements between consecutive layers' pools */
  const void* block_table;  /* int32 [n_seq][max_pages]; NULL for decode over an identity-layout pool */
  int max_pages;
  /* ABI v4: the 8-bit pools of the uniform quantized KV cache (vlm_attn_decode_paged_q8), layer 0 base, NULL = none.
   * When set, every decode step attends over them (the qkv epilogue still writes the new token to the bf16 pools; the
   * attention launch quantises it). */
  void *kpool8, *vpool8, *ksb, *vsb;
  /* ABI v5: 1 = the LAST layer of a stack deeper than 2 keeps attending over the bf16 pools - the reference's policy for
   * BATCHED generation with kv_bits (models/cache.py:8-21 should_quantize_kv_layer, generate/ar.py:842-858); 0 = every
   * layer attends over the 8-bit pools (generate_step's maybe_quantize_kv_cache, generate/common.py:170-181). */
  int q8_skip_last;
} vlm_kv_pool;

/* max_kv_size (RotatingKVCache, models/cache.py:442-625; make_prompt_cache cache.py:45-70 builds it with keep = 4): entry i
 * moves the cached token of sequence seq[i] at slot src_slot[i] to slot dst_slot[i] in EVERY layer (bf16 pools).  The source
 * and destination SETS of one call must be disjoint.  The host keeps the ring (which slot holds which token); the engine's
 * rule stays "write at slot = entries held, attend over entries held + 1" (csrc/kv_rotate.hip).  int32 device arrays [T]. */
int vlm_kv_move_tokens(void* kpool, void* vpool, size_t layer_stride, int n_layers, const void* seq, const void* src_slot,
                       const void* dst_slot, int T, const void* block_table, int max_pages, int Hkv, int D, void* stream);

/* KVCache.update_and_fetch (models/cache.py:345-367; BatchKVCache.update_and_fetch cache.py:1002-1025 calls it per row): the S
 * tokens of `keys` / `values` - bf16 [Hkv][S][D] views, strides in ELEMENTS (multiples of 8 for the keys, whose rows are read
 * as 16-byte pieces; base 16-byte aligned) - become the cached tokens slot0 .. slot0 + S - 1 of sequence `seq` in ONE layer:
 * kpool_layer / vpool_layer = that layer's pools (vlm_kv_pool.kpool + layer * layer_stride elements).  block_table NULL =
 * identity layout (page = seq * max_pages + index).  The pages must already belong to the sequence. */
int vlm_kv_append_tokens(void* kpool_layer, void* vpool_layer, const void* keys, const void* values, int S, long k_head_stride,
                         long k_tok_stride, long v_head_stride, long v_tok_stride, int seq, int slot0, const void* block_table,
                         int max_pages, int Hkv, int D, void* stream);

/* prefill over T tokens (all sequences concatenated).  h [T][hidden] holds the input
 * embeddings and is the residual stream (overwritten).  Workspaces are caller-owned:
 * xn [T][hidden], qkv [T][(Hq+2Hkv)*D], attn [T][Hq*D], act [T][inter].
 * last_rows int32 [n_last]: rows of h whose logits are wanted (the reference computes all L rows
 * and keeps [:, -1], ar.py:358); logits [n_last][vocab]; xlast [n_last][hidden] workspace. */
typedef struct vlm_prefill_args {
  void* h;
  int T;
  const void *pos_t, *pos_h, *pos_w; /* int32 [T] */
  const void *kv_seq, *kv_slot;      /* int32 [T] */
  const void* cu_seqlens;            /* int32 [nseg+1] */
  int nseg, total_qblocks;
  void *xn, *qkv, *attn, *act;
  const void* last_rows;
  int n_last;
  void *xlast, *logits;
  int rope_long; /* models with rope_long_from > 0: 1 = this call uses the LONG frequency table (the caller applies
                    SuScaledRoPE's per-call rule: max cache offset + tokens of the call > original_max) */
} vlm_prefill_args;

/* one decode step for B sequences (B in {1,2,4,8}); every buffer is device resident so the
 * step can be replayed from a graph: tok int32 [B] (in: token to feed, out: sampled token),
 * pos / ctx int32 [B] (rope position / keys in cache, advanced by the step), step int32 [1]. */
typedef struct vlm_decode_args {
  int B;
  void *tok, *pos, *ctx, *step;
  void *h, *qkv, *attn, *act, *logits, *logprobs, *scratch;
  void *part_o, *part_ml, *sample_ws;
  void* out_ring; /* int32 [ring_len][B] sampled-token history (may be NULL) */
  int ring_len, nsplit;
  float temperature, top_p, min_p;
  int top_k;
  unsigned seed;
  int flags; /* VLM_DECODE_* */
  const vlm_penalty_args* penalties; /* NULL: none.  Applied to the step's logits before sampling; the fed token is
                                        pushed to the history first (vlm_apply_logit_penalties). (host pointer, copied) */
} vlm_decode_args;

/* vlm_decode_args.flags */
#define VLM_DECODE_FUSED_TAIL 1 /* the step does not start with the embedding gather - h already holds embed[tok] (the
                                   caller gathers it once after the prefill, vlm_embed_gather) and the sampler tail leaves
                                   the next step's h behind: vlm_sample_greedy_advance at temperature 0; with a temperature
                                   the sampler's last launch (final pick) also does vlm_decode_advance and the gather.
                                   bf16 embedding tables only.  The host must not rewrite tok between steps. */
#define VLM_DECODE_ACT16 2      /* `act` holds 16 x intermediate_size elements (not B x): a step of 5..16 rows over bf16 weights
                                   may then hand the SwiGLU output to the down projection in the tiled layout of
                                   VLM_EPI_Y_TILED / VLM_EPI_X_TILED (rows padded to the MFMA tile's 16).  ABI v8. */

int vlm_llm_create(const vlm_llm_config* cfg, void** handle);          /* (host) */
int vlm_llm_destroy(void* handle);
int vlm_llm_set_layer(void* handle, int layer, const vlm_llm_layer* w);
int vlm_llm_set_globals(void* handle, const vlm_llm_globals* g);
int vlm_llm_set_kv(void* handle, const vlm_kv_pool* kv);
int vlm_llm_prefill(void* handle, const vlm_prefill_args* a, void* stream);
int vlm_llm_decode_step(void* handle, const vlm_decode_args* a, void* stream);
/* embeddings -> logits only (no sampling, no advance): language_model(y, cache=...) at L == 1 */
int vlm_llm_decode_forward(void* handle, const vlm_decode_args* a, void* stream);
/* capture vlm_llm_decode_step into a hipGraph owned by the handle / replay it */
int vlm_llm_decode_graph_build(void* handle, const vlm_decode_args* a, void* stream);
int vlm_llm_decode_graph_launch(void* handle, void* stream);
/* number of kernel launches in one decode step (for reporting) */
int vlm_llm_decode_launches(void* handle);

/* Tuning of the captured decode step (no effect on results; measured defaults in DESIGN.md).  Changing a value drops
 * the cached graphs of the handle.
 * (keys 0..5 and 10 were the round-2 / round-3 experiments - weight prefetch on a side branch, the fused MLP launch,
 *  launch-skipping masks, translation warm-up - which lost to the five-launch layer; they left the library in ABI v5 and
 *  live on as sources under scripts/rejected/ with their measurements in profiles/) */
#define VLM_TUNE_MFMA_GEMV 6        /* 1 (default): decode steps of 3..16 rows run their projections on the matrix cores
                                       (csrc/gemv_mfma.hip); 0: the v_dot2c GEMVs (rows in {4, 8}) - A/B knob */
#define VLM_TUNE_ATTN_PAGESPLIT 7   /* decode attention of steps with at most 64 (row, kv head) pairs: N > 0 (default 16) = the
                                       page-split form (vlm_attn_decode_paged_split) with up to N workgroups per pair (32 when
                                       the caller asked for a split, i.e. a long context); 0 = the round-2 forms (one workgroup
                                       per pair, or split-K partials merged in the o_proj prologue) - A/B knob */
#define VLM_TUNE_GEMV_VARIANT 8     /* A/B knob of the batch-1 GEMV launch shapes (bit 0: down projection with 6 rows per
                                       workgroup = one workgroup per CU at N = 1536; bit 1: 3 rows) */
#define VLM_TUNE_ATTN_MERGE 9       /* where the page-split attention of a ONE-row step is merged: 1 (default) = in the o_proj
                                       GEMV's prologue (partial-only attention launch, vlm_gemv_attn_out_bf16; needs bf16 Wo,
                                       Hq * D <= 2048, <= 16 splits), 0 = by the attention launch's last-arriving workgroup */
int vlm_llm_set_tuning(void* handle, int key, int value);
int vlm_llm_get_tuning(void* handle, int key);

/* The encoder-layer loop of the SigLIP / CLIP vision towers (idefics2/vision.py:141-187, llava_bunny/vision.py:139-200,
 * phi3_v/vision.py:117-175: x = x + out_proj(attention(LN1(x))); x = x + fc2(act(fc1(LN2(x)))), biases everywhere, no
 * rope, no mask) as one native call: n_layers x [LayerNorm, qkv GEMM + bias, varlen flash attention, out GEMM + bias +
 * residual, LayerNorm, fc1 GEMM + bias + activation, fc2 GEMM + bias + residual] enqueued back to back.
 * x [N][E] residual stream (in / out); workspaces xn [N][E], qkv [N][3 H head_dim], attn [N][H head_dim], mlp [N][MH];
 * wqkv rows [q | k | v] of H * head_dim each (head_dim = the kernel's width: 64 / 80 / 128 - narrower heads zero-padded
 * by the caller, wo's columns likewise); act_epilogue VLM_EPI_GELU_FAST / VLM_EPI_GELU_ERF / 0; cu_seqlens int32
 * [nseg + 1], total_qblocks = sum ceil(len / 128); uniform_segments: vlm_attn_prefill's placement hint. */
typedef struct vlm_enc_layer {
  const void *ln1_w, *ln1_b, *wqkv, *bqkv, *wo, *bo, *ln2_w, *ln2_b, *w1, *b1, *w2, *b2;
} vlm_enc_layer;
int vlm_encoder_forward(const vlm_enc_layer* layers /* host array */, int n_layers, void* x, void* xn, void* qkv, void* attn,
                        void* mlp, int N, int E, int H, int head_dim, int MH, float ln_eps, int act_epilogue,
                        const void* cu_seqlens, int nseg, int total_qblocks, float scale, int uniform_segments, void* stream);

typedef struct vlm_vit_config {
  int depth, embed_dim, n_heads, mlp_hidden, patch_k /* padded K of the patch GEMM */, merge /* 2 */, out_dim;
  float ln_eps;
  int qk_interleaved; /* 1: the q / k rows of every wqkv / bqkv are interleaved per head (vlm_gemm_bf16_rope2d) and
                         cos_tab / sin_tab of vlm_vit_args are ONE table (sin_tab == cos_tab + N * head_dim / 2): the 2-D
                         rope runs in the qkv GEMM epilogue instead of a pass of its own */
} vlm_vit_config;

typedef struct vlm_vit_block {
  const void *ln1_w, *ln1_b, *wqkv, *bqkv, *wproj, *bproj, *ln2_w, *ln2_b, *wfc1, *bfc1, *wfc2, *bfc2;
} vlm_vit_block;

typedef struct vlm_vit_globals {
  const void *wpatch /* [embed_dim][patch_k] */, *ln_q_w, *ln_q_b, *wm0, *bm0, *wm2, *bm2;
} vlm_vit_globals;

/* VisionModel.__call__ (vision.py:257-290): patches bf16 [N][patch_k] -> out bf16 [N/merge^2][out_dim].
 * Workspaces: x [N][E], xn [N][E], qkv [N][3E], attn [N][E], mlp [N][mlp_hidden], mrg [N/4][4E]. */
typedef struct vlm_vit_args {
  const void* patches;
  int N;
  const void *cos_tab, *sin_tab; /* fp32 [N][head_dim/2] */
  const void* cu_seqlens;
  int nseg, total_qblocks;
  void *x, *xn, *qkv, *attn, *mlp, *mrg, *out;
  int uniform_segments;          /* 1: all segments have the same length (vlm_attn_prefill placement hint) */
} vlm_vit_args;

int vlm_vit_create(const vlm_vit_config* cfg, void** handle);
int vlm_vit_destroy(void* handle);
int vlm_vit_set_block(void* handle, int i, const vlm_vit_block* w);
int vlm_vit_set_globals(void* handle, const vlm_vit_globals* g);
int vlm_vit_forward(void* handle, const vlm_vit_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VLM_HIP_H_ */
