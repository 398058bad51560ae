// Moving cached tokens inside the paged K/V pool for gfx950: what `max_kv_size` needs (reference RotatingKVCache,
// mlx_vlm/models/cache.py:442-625, built by make_prompt_cache cache.py:45-70 with keep = 4).  The reference keeps a ring of
// max_size entries and overwrites the oldest non-sink one in place; attention does not care in which order the keys sit
// (their rotary phase was applied when they were written), so the paged engine keeps its rule "the step's token is written at
// slot = entries held, the step attends over entries held + 1" and the HOST keeps the ring (models/cache.py::PagedSequence):
// before a step on a full window the newest entry (always in the last slot) is moved into the slot of the token that leaves,
// and once, after a prompt longer than the window, the survivors beyond the window are moved into the holes inside it.  One
// launch moves a list of tokens in every layer; a token is Hkv x D keys in the [page][Hkv][D/8][64][8] pool (D/8 16-byte
// pieces) and Hkv x D values in the [page][Hkv][D][64] pool (D 2-byte pieces 128 B apart) - kilobytes per step.
#include "common.cuh"
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

}  // namespace

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
