// Rotary embeddings + paged KV write for gfx950 (HBM-bound element kernels,
// 16-byte vector accesses, fp32 math, one bf16 rounding).
//
// vlm_rope2d_vision  replaces apply_rotary_pos_emb_vision
//     (reference mlx_vlm/models/qwen2_vl/vision.py:35-50,141-142)
// vlm_mrope_kvwrite  replaces MRoPERotaryEmbedding.apply_rotary - the fused
//     Metal kernel's numerics (reference mlx_vlm/models/rope_utils.py:567-651,
//     1243-1286; selector rope_utils.py:519-526) - together with
//     KVCache.update_and_fetch (reference mlx_vlm/models/cache.py:345-367),
//     writing into a PAGED cache instead of a contiguous, 256-step-grown one.
//
// Paged KV layout (ours; 64 tokens per page, sized for decode reads):
//   K pool: [page][Hkv][D/8][64][8]  - a wave reading one 8-wide d-chunk of the
//           64 keys of a page issues one contiguous 1 KiB load (lane = key)
//   V pool: [page][Hkv][D][64 slots] - transposed, key slots in the k-slot order of the decode P.V MFMA
//           (vlm_vslot), so a V^T operand fragment (d = lane&15, 8 key slots) is one 16-byte load
#include "common.hpp"
#include "../../include/vlm_hip.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4& u, float* v) {
  v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
  v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 o;
  o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
  return o;
}

// qkv [N][3][H][D] bf16 (row stride ld elements); cos/sin [N][D/2] fp32.
// out[d]     = T(x[d] c[d] - x[d+D/2] s[d]);  out[d+D/2] = T(x[d+D/2] c[d] + x[d] s[d]),  d < D/2
__global__ __launch_bounds__(256) void rope2d_vision_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ cs,
                                                            const float* __restrict__ sn, int N, int H, int D, int ld) {
  const int half = D >> 1, cph = half >> 3;  // 8-wide chunks per half head
  const long total = (long)N * 2 * H * cph;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % cph);
  long t = idx / cph;
  const int h = (int)(t % H); t /= H;
  const int which = (int)(t % 2);
  const int n = (int)(t / 2);
  bf16_t* base = qkv + (size_t)n * ld + ((size_t)which * H + h) * D + c * 8;
  const uint4 lo = *reinterpret_cast<const uint4*>(base), hi = *reinterpret_cast<const uint4*>(base + half);
  float a[8], b[8], co[8], si[8], oa[8], ob[8];
  unpack8(lo, a); unpack8(hi, b);
  const float4* cp = reinterpret_cast<const float4*>(cs + (size_t)n * half + c * 8);
  const float4* sp = reinterpret_cast<const float4*>(sn + (size_t)n * half + c * 8);
  const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
  co[0] = c0.x; co[1] = c0.y; co[2] = c0.z; co[3] = c0.w; co[4] = c1.x; co[5] = c1.y; co[6] = c1.z; co[7] = c1.w;
  si[0] = s0.x; si[1] = s0.y; si[2] = s0.z; si[3] = s0.w; si[4] = s1.x; si[5] = s1.y; si[6] = s1.z; si[7] = s1.w;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    oa[j] = a[j] * co[j] - b[j] * si[j];
    ob[j] = b[j] * co[j] + a[j] * si[j];
  }
  *reinterpret_cast<uint4*>(base) = pack8(oa);
  *reinterpret_cast<uint4*>(base + half) = pack8(ob);
}

// One thread = one 8-wide chunk of the low half of a q/k head (+ its partners
// in the high half), or one 8-wide chunk of a v head (copy to the cache).
__global__ __launch_bounds__(256) void mrope_kvwrite_kernel(
    bf16_t* __restrict__ qkv, int ld, int T, int Hq, int Hkv, int D, const int* __restrict__ pos_t,
    const int* __restrict__ pos_h, const int* __restrict__ pos_w, const float* __restrict__ inv_freq, int sec0, int sec1,
    const int* __restrict__ kv_seq, const int* __restrict__ kv_slot, const int* __restrict__ block_table, int max_pages,
    bf16_t* __restrict__ kpool, bf16_t* __restrict__ vpool, float qk_scale, int long_from) {
  const int half = D >> 1, cph = half >> 3;
  // SuScaledRoPE's per-call regime decided on the device (decode steps: T <= 64 rows, kv_slot = the rows' cache offsets):
  // inv_freq holds [2][D/2] (short, long) and the LONG row applies to EVERY row of the call when ANY row's offset has
  // reached original_max_position_embeddings (rope_utils.py:168-172) - the rule of the fused qkv kernels of gemv_*.hip
  if (long_from > 0) {
    bool any_long = false;
    for (int r = 0; r < T; ++r) any_long |= kv_slot[r] >= long_from;
    if (any_long) inv_freq += half;
  }
  const int rot_items = (Hq + Hkv) * cph;      // rotary chunks per token
  const int v_items = Hkv * (D >> 3);          // v copy chunks per token
  const int per_tok = rot_items + v_items;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)T * per_tok) return;
  const int tok = (int)(idx / per_tok);
  const int it = (int)(idx % per_tok);
  bf16_t* row = qkv + (size_t)tok * ld;

  long page = -1;
  int within = 0;
  if (kpool) {
    const int seq = kv_seq ? kv_seq[tok] : tok;
    const int slot = kv_slot[tok];
    page = block_table[(size_t)seq * max_pages + (slot >> 6)];
    within = slot & 63;
  }

  if (it < rot_items) {
    const int head = it / cph, c = it % cph;
    bf16_t* base = row + (size_t)head * D + c * 8;
    float a[8], b[8], oa[8], ob[8];
    unpack8(*reinterpret_cast<const uint4*>(base), a);
    unpack8(*reinterpret_cast<const uint4*>(base + half), b);
    const float pt = (float)pos_t[tok], ph = (float)pos_h[tok], pw = (float)pos_w[tok];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = c * 8 + j;
      const float p = f < sec0 ? pt : (f < sec0 + sec1 ? ph : pw);
      const float ang = p * inv_freq[f];
      float s, co;
      sincosf(ang, &s, &co);
      // SuScaledRoPE multiplies q / k by T(scale) first (a typed op: one rounding; exact no-op at scale 1)
      const float za = rbf(a[j] * qk_scale), zb = rbf(b[j] * qk_scale);
      oa[j] = za * co - zb * s;
      ob[j] = zb * co + za * s;
    }
    const uint4 lo = pack8(oa), hi = pack8(ob);
    *reinterpret_cast<uint4*>(base) = lo;
    *reinterpret_cast<uint4*>(base + half) = hi;
    if (head >= Hq && kpool) {
      const int g = head - Hq;
      const size_t kb = ((size_t)page * Hkv + g) * (size_t)(D >> 3);
      *reinterpret_cast<uint4*>(kpool + ((kb + c) * 64 + within) * 8) = lo;
      *reinterpret_cast<uint4*>(kpool + ((kb + c + cph) * 64 + within) * 8) = hi;
    }
  } else if (vpool) {
    const int vi = it - rot_items;
    const int g = vi / (D >> 3), c = vi % (D >> 3);
    const uint4 v = *reinterpret_cast<const uint4*>(row + (size_t)(Hq + Hkv + g) * D + c * 8);
    bf16_t* vb = vpool + (((size_t)page * Hkv + g) * D + c * 8) * 64 + vlm_vslot(within);   // [D][64 slots]
    vb[0 * 64] = (bf16_t)(v.x & 0xffffu); vb[1 * 64] = (bf16_t)(v.x >> 16);
    vb[2 * 64] = (bf16_t)(v.y & 0xffffu); vb[3 * 64] = (bf16_t)(v.y >> 16);
    vb[4 * 64] = (bf16_t)(v.z & 0xffffu); vb[5 * 64] = (bf16_t)(v.z >> 16);
    vb[6 * 64] = (bf16_t)(v.w & 0xffffu); vb[7 * 64] = (bf16_t)(v.w >> 16);
  }
}

// the fetch half of KVCache.update_and_fetch (reference cache.py:345-367: `return self.keys[..., :offset, :], ...`) for a
// PAGED cache: token t's cached (already rotated) k and v rows -> the k / v columns of row t of a token-major qkv buffer
__global__ __launch_bounds__(256) void kv_gather_kernel(bf16_t* __restrict__ qkv, int ld, int T, int Hq, int Hkv, int D,
                                                        const int* __restrict__ kv_seq, const int* __restrict__ kv_slot,
                                                        const int* __restrict__ block_table, int max_pages,
                                                        const bf16_t* __restrict__ kpool, const bf16_t* __restrict__ vpool) {
  const int cpd = D >> 3, per_tok = 2 * Hkv * cpd;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)T * per_tok) return;
  const int tok = (int)(idx / per_tok);
  int it = (int)(idx % per_tok);
  const bool is_v = it >= Hkv * cpd;
  if (is_v) it -= Hkv * cpd;
  const int g = it / cpd, c = it % cpd;
  const int seq = kv_seq ? kv_seq[tok] : tok, slot = kv_slot[tok];
  const size_t page = (size_t)block_table[(size_t)seq * max_pages + (slot >> 6)];
  const int within = slot & 63;
  bf16_t* row = qkv + (size_t)tok * ld;
  if (!is_v) {
    const uint4 k = *reinterpret_cast<const uint4*>(kpool + (((page * Hkv + g) * (size_t)cpd + c) * 64 + within) * 8);
    *reinterpret_cast<uint4*>(row + (size_t)(Hq + g) * D + c * 8) = k;
  } else {
    const bf16_t* vb = vpool + ((page * Hkv + g) * (size_t)D + c * 8) * 64 + vlm_vslot(within);
    uint4 v;
    v.x = (uint32_t)vb[0 * 64] | ((uint32_t)vb[1 * 64] << 16);
    v.y = (uint32_t)vb[2 * 64] | ((uint32_t)vb[3 * 64] << 16);
    v.z = (uint32_t)vb[4 * 64] | ((uint32_t)vb[5 * 64] << 16);
    v.w = (uint32_t)vb[6 * 64] | ((uint32_t)vb[7 * 64] << 16);
    *reinterpret_cast<uint4*>(row + (size_t)(Hq + Hkv + g) * D + c * 8) = v;
  }
}

}  // namespace

extern "C" int vlm_kv_gather(void* qkv, int ld, int T, int Hq, int Hkv, int D, const void* kv_seq, const void* kv_slot,
                             const void* block_table, int max_pages, const void* kpool, const void* vpool, void* stream) {
  if (!qkv || !kv_slot || !block_table || !kpool || !vpool || T < 0 || Hq < 0 || Hkv <= 0 || max_pages <= 0) return VLM_ERR_ARG;
  if (D % 8 != 0 || ld % 8 != 0) return VLM_ERR_SHAPE;
  if (T == 0) return VLM_OK;
  const long total = (long)T * 2 * Hkv * (D / 8);
  hipLaunchKernelGGL(kv_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)qkv,
                     ld, T, Hq, Hkv, D, (const int*)kv_seq, (const int*)kv_slot, (const int*)block_table, max_pages,
                     (const bf16_t*)kpool, (const bf16_t*)vpool);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_rope2d_vision(void* qkv, const void* cos_tab, const void* sin_tab, int N, int H, int D, int ld,
                                 void* stream) {
  if (!qkv || !cos_tab || !sin_tab || N < 0 || H <= 0 || D <= 0) return VLM_ERR_ARG;
  if (D % 16 != 0 || ld % 8 != 0) return VLM_ERR_SHAPE;
  if (N == 0) return VLM_OK;
  const long total = (long)N * 2 * H * (D / 16);
  hipLaunchKernelGGL(rope2d_vision_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)qkv, (const float*)cos_tab, (const float*)sin_tab, N, H, D, ld);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_mrope_kvwrite(void* qkv, int ld, int T, int Hq, int Hkv, int D, const void* pos_t, const void* pos_h,
                                 const void* pos_w, const void* inv_freq, int sec0, int sec1, const void* kv_seq,
                                 const void* kv_slot, const void* block_table, int max_pages, void* kpool, void* vpool,
                                 void* stream) {
  return vlm_mrope_kvwrite_scaled(qkv, ld, T, Hq, Hkv, D, pos_t, pos_h, pos_w, inv_freq, sec0, sec1, kv_seq, kv_slot, block_table,
                              max_pages, kpool, vpool, 1.f, stream);
}

extern "C" int vlm_mrope_kvwrite_scaled(void* qkv, int ld, int T, int Hq, int Hkv, int D, const void* pos_t, const void* pos_h,
                                        const void* pos_w, const void* inv_freq, int sec0, int sec1, const void* kv_seq,
                                        const void* kv_slot, const void* block_table, int max_pages, void* kpool, void* vpool,
                                        float qk_scale, void* stream) {
  if (!qkv || !pos_t || !pos_h || !pos_w || !inv_freq || T < 0 || Hq <= 0 || Hkv <= 0) return VLM_ERR_ARG;
  if ((kpool != nullptr) != (vpool != nullptr)) return VLM_ERR_ARG;
  if (kpool && (!kv_slot || !block_table || max_pages <= 0)) return VLM_ERR_ARG;
  if (D % 16 != 0 || ld % 8 != 0) return VLM_ERR_SHAPE;
  if (T == 0) return VLM_OK;
  const long total = (long)T * ((Hq + Hkv) * (D / 16) + Hkv * (D / 8));
  hipLaunchKernelGGL(mrope_kvwrite_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)qkv, ld, T, Hq, Hkv, D, (const int*)pos_t, (const int*)pos_h, (const int*)pos_w,
                     (const float*)inv_freq, sec0, sec1, (const int*)kv_seq, (const int*)kv_slot,
                     (const int*)block_table, max_pages, (bf16_t*)kpool, (bf16_t*)vpool, qk_scale, 0);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// decode form (library-internal, csrc/engine.hip's wide steps): row b = one decoded token at text position pos[b] (all three
// M-RoPE axes equal), written at slot[b] of block-table row b; long_from > 0: inv_freq = [2][D/2] and the call-wide regime
// is decided from the slots on the device, so a captured wide step crosses the limit without host help
int vlm_mrope_kvwrite_decode(void* qkv, int ld, int B, int Hq, int Hkv, int D, const void* pos, const void* inv_freq, int sec0,
                             int sec1, const void* slot, const void* block_table, int max_pages, void* kpool, void* vpool,
                             float qk_scale, int long_from, void* stream) {
  if (!qkv || !pos || !inv_freq || !slot || !block_table || !kpool || !vpool || B <= 0 || B > 64 || Hq <= 0 || Hkv <= 0 || max_pages <= 0)
    return VLM_ERR_ARG;
  if (D % 16 != 0 || ld % 8 != 0) return VLM_ERR_SHAPE;
  const long total = (long)B * ((Hq + Hkv) * (D / 16) + Hkv * (D / 8));
  hipLaunchKernelGGL(mrope_kvwrite_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)qkv, ld, B, Hq, Hkv, D, (const int*)pos, (const int*)pos, (const int*)pos, (const float*)inv_freq,
                     sec0, sec1, (const int*)nullptr, (const int*)slot, (const int*)block_table, max_pages, (bf16_t*)kpool,
                     (bf16_t*)vpool, qk_scale, long_from);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
