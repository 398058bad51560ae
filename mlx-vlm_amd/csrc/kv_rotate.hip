// Moving cached tokens inside the paged K/V pool for gfx950: what `max_kv_size` needs (reference RotatingKVCache,
// mlx_vlm/models/cache.py:442-625, built by make_prompt_cache cache.py:45-70 with keep = 4).  The reference keeps a ring of
// max_size entries and overwrites the oldest non-sink one in place; attention does not care in which order the keys sit
// (their rotary phase was applied when they were written), so the paged engine keeps its rule "the step's token is written at
// slot = entries held, the step attends over entries held + 1" and the HOST keeps the ring (models/cache.py::PagedSequence):
// before a step on a full window the newest entry (always in the last slot) is moved into the slot of the token that leaves,
// and once, after a prompt longer than the window, the survivors beyond the window are moved into the holes inside it.  One
// launch moves a list of tokens in every layer; a token is Hkv x D keys in the [page][Hkv][D/8][64][8] pool (D/8 16-byte
// pieces) and Hkv x D values in the [page][Hkv][D][64] pool (D 2-byte pieces 128 B apart) - kilobytes per step.
#include "common.hpp"
#include "internal.h"
#include "../../include/vlm_hip.h"

namespace {

constexpr int PAGE = 64;

__global__ __launch_bounds__(128) void kv_move_tokens_kernel(bf16_t* __restrict__ kpool, bf16_t* __restrict__ vpool,
                                                             size_t layer_stride, const int* __restrict__ seq,
                                                             const int* __restrict__ src_slot, const int* __restrict__ dst_slot,
                                                             const int* __restrict__ block_table, int max_pages, int Hkv, int D) {
  const int i = blockIdx.x, g = blockIdx.y, layer = blockIdx.z, tid = threadIdx.x;
  const int sq = seq[i], s = src_slot[i], d = dst_slot[i];
  if (s == d) return;
  const size_t ps = block_table ? (size_t)block_table[(size_t)sq * max_pages + (s >> 6)] : (size_t)sq * max_pages + (s >> 6);
  const size_t pd = block_table ? (size_t)block_table[(size_t)sq * max_pages + (d >> 6)] : (size_t)sq * max_pages + (d >> 6);
  const size_t lo = (size_t)layer * layer_stride;
  const int ws = s & 63, wd = d & 63;
  const bf16_t* ks = kpool + lo + (ps * Hkv + g) * (size_t)D * PAGE;
  bf16_t* kd = kpool + lo + (pd * Hkv + g) * (size_t)D * PAGE;
  for (int c = tid; c < (D >> 3); c += 128)
    *reinterpret_cast<uint4*>(kd + ((size_t)c * PAGE + wd) * 8) = *reinterpret_cast<const uint4*>(ks + ((size_t)c * PAGE + ws) * 8);
  const bf16_t* vs = vpool + lo + (ps * Hkv + g) * (size_t)D * PAGE + vlm_vslot(ws);
  bf16_t* vd = vpool + lo + (pd * Hkv + g) * (size_t)D * PAGE + vlm_vslot(wd);
  for (int e = tid; e < D; e += 128) vd[(size_t)e * PAGE] = vs[(size_t)e * PAGE];
}

// KVCache.update_and_fetch of the reference (cache.py:345-367) on the paged pool: tokens 0..S-1 of `keys` / `values`
// ([Hkv][S][D] views with element strides) become the cached tokens slot0 .. slot0 + S - 1 of one sequence in ONE layer.
// One workgroup per (token, kv head): D / 8 16-byte K pieces, D 2-byte V pieces 128 B apart (the pool layouts of the header).
__global__ __launch_bounds__(128) void kv_append_tokens_kernel(bf16_t* __restrict__ kpool, bf16_t* __restrict__ vpool,
                                                               const bf16_t* __restrict__ keys, const bf16_t* __restrict__ values,
                                                               long k_head_stride, long k_tok_stride, long v_head_stride,
                                                               long v_tok_stride, int seq, int slot0,
                                                               const int* __restrict__ block_table, int max_pages, int Hkv, int D) {
  const int t = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int slot = slot0 + t, w = slot & 63;
  const size_t page = block_table ? (size_t)block_table[(size_t)seq * max_pages + (slot >> 6)] : (size_t)seq * max_pages + (slot >> 6);
  const bf16_t* ks = keys + (size_t)g * k_head_stride + (size_t)t * k_tok_stride;
  bf16_t* kd = kpool + (page * Hkv + g) * (size_t)D * PAGE;
  for (int c = tid; c < (D >> 3); c += 128)
    *reinterpret_cast<uint4*>(kd + ((size_t)c * PAGE + w) * 8) = *reinterpret_cast<const uint4*>(ks + 8 * c);
  const bf16_t* vs = values + (size_t)g * v_head_stride + (size_t)t * v_tok_stride;
  bf16_t* vd = vpool + (page * Hkv + g) * (size_t)D * PAGE + vlm_vslot(w);
  for (int e = tid; e < D; e += 128) vd[(size_t)e * PAGE] = vs[e];
}

}  // namespace

extern "C" int vlm_kv_append_tokens(void* kpool_layer, void* vpool_layer, const void* keys, const void* values, int S,
                                    long k_head_stride, long k_tok_stride, long v_head_stride, long v_tok_stride, int seq, int slot0,
                                    const void* block_table, int max_pages, int Hkv, int D, void* stream) {
  if (!kpool_layer || !vpool_layer || !keys || !values || S < 0 || seq < 0 || slot0 < 0 || max_pages <= 0 || Hkv <= 0) return VLM_ERR_ARG;
  if (D <= 0 || D % 8 || k_head_stride % 8 || k_tok_stride % 8 || ((size_t)keys & 15)) return VLM_ERR_SHAPE;
  if ((slot0 + S + 63) / 64 > max_pages) return VLM_ERR_SHAPE;
  if (S == 0) return VLM_OK;
  if (Hkv > 65535) return VLM_ERR_SHAPE;
  hipLaunchKernelGGL(kv_append_tokens_kernel, dim3(S, Hkv), dim3(128), 0, (hipStream_t)stream, (bf16_t*)kpool_layer,
                     (bf16_t*)vpool_layer, (const bf16_t*)keys, (const bf16_t*)values, k_head_stride, k_tok_stride, v_head_stride,
                     v_tok_stride, seq, slot0, (const int*)block_table, max_pages, Hkv, D);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_kv_move_tokens(void* kpool, void* vpool, size_t layer_stride, int n_layers, const void* seq, const void* src_slot,
                                  const void* dst_slot, int T, const void* block_table, int max_pages, int Hkv, int D, void* stream) {
  if (!kpool || !vpool || !seq || !src_slot || !dst_slot || n_layers <= 0 || T < 0 || max_pages <= 0 || Hkv <= 0) return VLM_ERR_ARG;
  if (D <= 0 || D % 8) return VLM_ERR_SHAPE;
  if (T == 0) return VLM_OK;
  if (T > 65535 * 1024 || Hkv > 65535 || n_layers > 65535) return VLM_ERR_SHAPE;
  hipLaunchKernelGGL(kv_move_tokens_kernel, dim3(T, Hkv, n_layers), dim3(128), 0, (hipStream_t)stream, (bf16_t*)kpool,
                     (bf16_t*)vpool, layer_stride, (const int*)seq, (const int*)src_slot, (const int*)dst_slot,
                     (const int*)block_table, max_pages, Hkv, D);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
