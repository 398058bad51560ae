// Row gather / scatter / cast kernels for gfx950 (HBM-bound, 16-byte vectors).
//
// vlm_embed_gather        nn.Embedding (reference mlx_vlm/models/qwen2_vl/language.py:164,179)
// vlm_scatter_image_rows  Model.merge_input_ids_with_image_features
//                         (reference mlx_vlm/models/qwen2_vl/qwen2_vl.py:78-148): the cumsum /
//                         gather / where there collapses to "row dst_rows[i] of the embeddings :=
//                         image feature row i" once the image-token positions are known on the host
// vlm_cast_f32_bf16_pad   pixel_values.astype(weight dtype) (qwen2_vl.py:44-45) fused with the
//                         zero padding of K = 1176 -> 1216 the patch GEMM wants
// vlm_decode_advance      per-step bookkeeping of the decode loop: KVCache.offset += 1
//                         (reference mlx_vlm/models/cache.py:362) and pos = offset + rope_delta
//                         (language.py:476-509) kept in device memory so a step can be graph-replayed
#include "common.hpp"
#include "../../include/vlm_hip.h"

namespace {

__global__ __launch_bounds__(256) void embed_gather_kernel(const int* __restrict__ ids, const bf16_t* __restrict__ table,
                                                           bf16_t* __restrict__ out, int T, int D, int ldo, int vocab) {
  const int cpr = D >> 3;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)T * cpr) return;
  const int t = (int)(idx / cpr), c = (int)(idx % cpr);
  int id = ids[t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  reinterpret_cast<uint4*>(out + (size_t)t * ldo)[c] = reinterpret_cast<const uint4*>(table + (size_t)id * D)[c];
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ dst_rows,
                                                           bf16_t* __restrict__ dst, int n, int D, int lds_, int ldd) {
  const int cpr = D >> 3;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n * cpr) return;
  const int r = (int)(idx / cpr), c = (int)(idx % cpr);
  reinterpret_cast<uint4*>(dst + (size_t)dst_rows[r] * ldd)[c] = reinterpret_cast<const uint4*>(src + (size_t)r * lds_)[c];
}

__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int rows,
                                                       int cols, int ld_src, int ld_dst) {
  const int cpr = ld_dst >> 3;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * cpr) return;
  const int r = (int)(idx / cpr), c = (int)(idx % cpr);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = c * 8 + j;
    v[j] = col < cols ? src[(size_t)r * ld_src + col] : 0.f;
  }
  uint4 o;
  o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
  reinterpret_cast<uint4*>(dst + (size_t)r * ld_dst)[c] = o;
}

__global__ void decode_advance_kernel(int* __restrict__ ctx, int* __restrict__ pos, const int* __restrict__ tok,
                                      int* __restrict__ out_ring, int ring_len, int* __restrict__ step, int B) {
  const int b = threadIdx.x;
  const int s = *step;
  if (b < B) {
    ctx[b] += 1;
    pos[b] += 1;
    if (out_ring) out_ring[(size_t)(s % ring_len) * B + b] = tok[b];
  }
  __syncthreads();
  if (b == 0) *step = s + 1;
}

}  // namespace

extern "C" int vlm_embed_gather(const void* ids, const void* table, void* out, int T, int D, int ldo, int vocab,
                                void* stream) {
  if (!ids || !table || !out || T < 0 || D <= 0 || vocab <= 0) return VLM_ERR_ARG;
  if (D % 8 || ldo % 8) return VLM_ERR_SHAPE;
  if (T == 0) return VLM_OK;
  const long total = (long)T * (D / 8);
  hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const int*)ids, (const bf16_t*)table, (bf16_t*)out, T, D, ldo, vocab);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_scatter_image_rows(const void* src, const void* dst_rows, void* dst, int n, int D, int ld_src,
                                      int ld_dst, void* stream) {
  if (!src || !dst_rows || !dst || n < 0 || D <= 0) return VLM_ERR_ARG;
  if (D % 8 || ld_src % 8 || ld_dst % 8) return VLM_ERR_SHAPE;
  if (n == 0) return VLM_OK;
  const long total = (long)n * (D / 8);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (const int*)dst_rows, (bf16_t*)dst, n, D, ld_src, ld_dst);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_cast_f32_bf16_pad(const void* src, void* dst, int rows, int cols, int ld_src, int ld_dst,
                                     void* stream) {
  if (!src || !dst || rows < 0 || cols <= 0 || ld_dst < cols) return VLM_ERR_ARG;
  if (ld_dst % 8) return VLM_ERR_SHAPE;
  if (rows == 0) return VLM_OK;
  const long total = (long)rows * (ld_dst / 8);
  hipLaunchKernelGGL(cast_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)src, (bf16_t*)dst, rows, cols, ld_src, ld_dst);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_decode_advance(void* ctx, void* pos, const void* tok, void* out_ring, int ring_len, void* step, int B,
                                  void* stream) {
  if (!ctx || !pos || !tok || !step || B <= 0 || B > 256) return VLM_ERR_ARG;
  if (out_ring && ring_len <= 0) return VLM_ERR_ARG;
  hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (int*)ctx, (int*)pos,
                     (const int*)tok, (int*)out_ring, ring_len, (int*)step, B);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
