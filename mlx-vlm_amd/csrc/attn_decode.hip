// Per-step decode attention over the paged KV cache for gfx950 (HBM/L2-bound,
// tiny): split-K over the context, GQA group processed together so K/V pages
// are read once for all q heads that share them.
//
// Replaces mx.fast.scaled_dot_product_attention for L == 1
//   (reference mlx_vlm/models/base.py:366-373 called from
//    mlx_vlm/models/qwen2_vl/language.py:115-118; GQA n_heads / n_kv_heads from
//    language.py:44-49), mask=None (base.py:214-228: N == 1 -> no mask).
//
// Work decomposition: workgroup = (sequence b, kv head g, split s); its 4 waves
// take pages s*pps + w, +4, ...  Inside a page a lane IS a key for the Q.K pass
// (K pool is [page][Hkv][D/8][64][8]: one contiguous 1 KiB load per 8-wide
// d-chunk, no cross-lane reduction for the dot products), the per-head softmax
// statistics are wavefront shuffles, and the P.V pass re-maps lanes to
// (key%4, 8-wide d-chunk) so V rows are read as 16-byte vectors; fp32 throughout.
// Partials (m, l, O) go to a small fp32 workspace; vlm_attn_decode_combine
// merges the splits and writes bf16.
#include "common.cuh"
#include "../../include/vlm_hip.h"

namespace {

constexpr int HD = 128;   // head_dim supported by the decode path
constexpr int PAGE = 64;

template <int G>
__global__ __launch_bounds__(256) void attn_decode_kernel(
    const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kpool, const bf16_t* __restrict__ vpool,
    const int* __restrict__ block_table, int max_pages, const int* __restrict__ kv_len, int kv_len_add, int Hq, int Hkv,
    float scale, int nsplit, float* __restrict__ part_o, float* __restrict__ part_ml) {
  __shared__ __attribute__((aligned(16))) float qs[G][HD];
  __shared__ __attribute__((aligned(16))) float red_o[4][G][HD];
  __shared__ float red_m[4][G], red_l[4][G];

  const int b = blockIdx.x / Hkv, g = blockIdx.x % Hkv, s = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = kv_len[b] + kv_len_add;
  const int npages = (len + PAGE - 1) / PAGE;
  const int pps = (npages + nsplit - 1) / nsplit;
  const int p_begin = s * pps, p_end = min(npages, p_begin + pps);

  // q (pre-scaled) -> LDS as fp32
  for (int i = tid; i < G * HD; i += 256) {
    const int gg = i / HD, d = i % HD;
    qs[gg][d] = bf2f(q[(size_t)b * ldq + (size_t)(g * G + gg) * HD + d]) * scale;
  }
  __syncthreads();

  float m[G], l[G], o[G][8];
#pragma unroll
  for (int gg = 0; gg < G; ++gg) {
    m[gg] = -INFINITY; l[gg] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[gg][j] = 0.f;
  }
  const int dchunk = lane & 15, ksub = lane >> 4;  // P.V mapping

  for (int pi = p_begin + wave; pi < p_end; pi += 4) {
    const long page = block_table[(size_t)b * max_pages + pi];
    const bf16_t* kp = kpool + ((size_t)page * Hkv + g) * (size_t)(HD / 8) * PAGE * 8;
    const bf16_t* vp = vpool + ((size_t)page * Hkv + g) * (size_t)PAGE * HD;
    const int key = pi * PAGE + lane;
    float sc[G];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) sc[gg] = 0.f;
#pragma unroll 4
    for (int c = 0; c < HD / 8; ++c) {
      const uint4 ku = *reinterpret_cast<const uint4*>(kp + ((size_t)c * PAGE + lane) * 8);
      const float kf[8] = {bf_lo(ku.x), bf_hi(ku.x), bf_lo(ku.y), bf_hi(ku.y), bf_lo(ku.z), bf_hi(ku.z), bf_lo(ku.w), bf_hi(ku.w)};
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const float4 q0 = *reinterpret_cast<const float4*>(&qs[gg][c * 8]);
        const float4 q1 = *reinterpret_cast<const float4*>(&qs[gg][c * 8 + 4]);
        sc[gg] += kf[0] * q0.x + kf[1] * q0.y + kf[2] * q0.z + kf[3] * q0.w + kf[4] * q1.x + kf[5] * q1.y + kf[6] * q1.z + kf[7] * q1.w;
      }
    }
    float p[G];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      const float sv = key < len ? sc[gg] : -INFINITY;
      const float mn = fmaxf(m[gg], wave_max(sv));   // page has >= 1 valid key, so mn is finite
      const float alpha = __expf(m[gg] - mn);
      p[gg] = __expf(sv - mn);
      l[gg] = l[gg] * alpha + wave_sum(p[gg]);
      m[gg] = mn;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[gg][j] *= alpha;
    }
    // P.V : lane -> (key = 4*jj + ksub, d = dchunk*8 .. +8)
#pragma unroll 4
    for (int jj = 0; jj < PAGE / 4; ++jj) {
      const int kk = jj * 4 + ksub;
      uint4 vu = *reinterpret_cast<const uint4*>(vp + (size_t)kk * HD + dchunk * 8);
      if (pi * PAGE + kk >= len) vu = make_uint4(0, 0, 0, 0);   // never-written slots: 0 * garbage must stay 0
      const float vf[8] = {bf_lo(vu.x), bf_hi(vu.x), bf_lo(vu.y), bf_hi(vu.y), bf_lo(vu.z), bf_hi(vu.z), bf_lo(vu.w), bf_hi(vu.w)};
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const float pj = __shfl(p[gg], kk, 64);   // p == 0 for keys beyond len
#pragma unroll
        for (int j = 0; j < 8; ++j) o[gg][j] += pj * vf[j];
      }
    }
  }

  // reduce the 4 key-subsets inside the wave, then the 4 waves through LDS
#pragma unroll
  for (int gg = 0; gg < G; ++gg)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = o[gg][j];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      o[gg][j] = v;
    }
  if (lane < 16) {
#pragma unroll
    for (int gg = 0; gg < G; ++gg)
#pragma unroll
      for (int j = 0; j < 8; ++j) red_o[wave][gg][lane * 8 + j] = o[gg][j];
  }
  if (lane == 0) {
#pragma unroll
    for (int gg = 0; gg < G; ++gg) { red_m[wave][gg] = m[gg]; red_l[wave][gg] = l[gg]; }
  }
  __syncthreads();
  for (int i = tid; i < G * HD; i += 256) {
    const int gg = i / HD, d = i % HD;
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) mm = fmaxf(mm, red_m[w][gg]);
    float acc = 0.f, ll = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = red_m[w][gg] == -INFINITY ? 0.f : __expf(red_m[w][gg] - mm);
      acc += f * red_o[w][gg][d];
      ll += f * red_l[w][gg];
    }
    const size_t hidx = ((size_t)b * Hq + (g * G + gg)) * nsplit + s;
    part_o[hidx * HD + d] = acc;
    if (d == 0) { part_ml[hidx * 2] = mm; part_ml[hidx * 2 + 1] = ll; }
  }
}

__global__ __launch_bounds__(128) void attn_decode_combine_kernel(const float* __restrict__ part_o,
                                                                  const float* __restrict__ part_ml, int nsplit,
                                                                  bf16_t* __restrict__ out, int ldo, int Hq) {
  const int bh = blockIdx.x, b = bh / Hq, h = bh % Hq, d = threadIdx.x;
  float mm = -INFINITY;
  for (int s = 0; s < nsplit; ++s) mm = fmaxf(mm, part_ml[((size_t)bh * nsplit + s) * 2]);
  float acc = 0.f, ll = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float ms = part_ml[((size_t)bh * nsplit + s) * 2];
    const float f = ms == -INFINITY ? 0.f : __expf(ms - mm);
    acc += f * part_o[((size_t)bh * nsplit + s) * HD + d];
    ll += f * part_ml[((size_t)bh * nsplit + s) * 2 + 1];
  }
  out[(size_t)b * ldo + (size_t)h * HD + d] = f2bf(acc / ll);
}

}  // namespace

extern "C" int vlm_attn_decode_paged(const void* q, int ldq, const void* kpool, const void* vpool,
                                     const void* block_table, int max_pages, const void* kv_len, int kv_len_add, int B,
                                     int Hq, int Hkv, int D, float scale, int nsplit, void* part_o, void* part_ml,
                                     void* out, int ldo, void* stream) {
  if (!q || !kpool || !vpool || !block_table || !kv_len || !part_o || !part_ml || !out) return VLM_ERR_ARG;
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || nsplit <= 0 || Hq % Hkv != 0) return VLM_ERR_ARG;
  if (D != HD) return VLM_ERR_SHAPE;
  const int G = Hq / Hkv;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(B * Hkv, nsplit), block(256);
#define GO(GV)                                                                                                         \
  hipLaunchKernelGGL((attn_decode_kernel<GV>), grid, block, 0, st, (const bf16_t*)q, ldq, (const bf16_t*)kpool,         \
                     (const bf16_t*)vpool, (const int*)block_table, max_pages, (const int*)kv_len, kv_len_add, Hq, Hkv, \
                     scale, nsplit, (float*)part_o, (float*)part_ml)
  switch (G) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 3: GO(3); break;
    case 4: GO(4); break;
    case 5: GO(5); break;
    case 6: GO(6); break;
    case 7: GO(7); break;
    case 8: GO(8); break;
    default: return VLM_ERR_SHAPE;
  }
#undef GO
  VLM_CHECK_LAUNCH();
  hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(B * Hq), dim3(HD), 0, st, (const float*)part_o,
                     (const float*)part_ml, nsplit, (bf16_t*)out, ldo, Hq);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
