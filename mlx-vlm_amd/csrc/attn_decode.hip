// Per-step decode attention over the paged KV cache for gfx950 (HBM/L2-bound,
// tiny): split-K over the context, GQA group processed together so K/V pages
// are read once for all q heads that share them.
//
// Replaces mx.fast.scaled_dot_product_attention for L == 1
//   (reference mlx_vlm/models/base.py:366-373 called from
//    mlx_vlm/models/qwen2_vl/language.py:115-118; GQA n_heads / n_kv_heads from
//    language.py:44-49), mask=None (base.py:214-228: N == 1 -> no mask).
//
// Work decomposition: workgroup = (sequence b, kv head g, split s) walks pages
// s, s + nsplit, ...; all 256 threads share ONE page at a time and every K and V
// load of the page is issued before any arithmetic, so a page costs one memory
// round trip (the kernel is latency-, not bandwidth-bound: 0.5 MB per layer).
// Inside a page a lane IS a key for the Q.K pass (K pool is
// [page][Hkv][D/8][64][8]: one contiguous 1 KiB load per 8-wide d-chunk, no
// cross-lane reduction for the dot products; the 4 waves split the 128 dims and
// meet in LDS), the per-head softmax statistics are wavefront shuffles, and the
// P.V pass re-maps lanes to (key%4, 8-wide d-chunk) so V rows are 16-byte loads;
// fp32 throughout.  Partials (m, l, O) go to a small fp32 workspace; the merge of
// the splits is fused into the o_proj GEMV prologue (vlm_gemv_attn_out) or done by
// the combine kernel below when the caller asks for the bf16 output.
#include "common.cuh"
#include "../../include/vlm_hip.h"

namespace {

constexpr int HD = 128;   // head_dim supported by the decode path
constexpr int PAGE = 64;

// workgroup = (sequence b, kv head g, split s); pages s, s + nsplit, ...  All 256 threads work on ONE page
// at a time so a page costs a single memory round trip: every K and V load of the page is issued up front.
//   Q.K : lane = key, wave w covers d-chunks [4w, 4w+4) (32 of the 128 dims); partial scores meet in LDS
//   P.V : wave w covers keys [16w, 16w+16); lane -> (key%4 subset, 8-wide d-chunk), 16-byte V loads
template <int G>
__global__ __launch_bounds__(256) void attn_decode_kernel(
    const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kpool, const bf16_t* __restrict__ vpool,
    const int* __restrict__ block_table, int max_pages, const int* __restrict__ kv_len, int kv_len_add, int Hq, int Hkv,
    float scale, int nsplit, float* __restrict__ part_o, float* __restrict__ part_ml) {
  __shared__ __attribute__((aligned(16))) float qs[G][HD];
  __shared__ __attribute__((aligned(16))) float sc_part[4][G][PAGE];
  __shared__ __attribute__((aligned(16))) float red_o[4][G][HD];

  const int b = blockIdx.x / Hkv, g = blockIdx.x % Hkv, s = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = kv_len[b] + kv_len_add;
  const int npages = (len + PAGE - 1) / PAGE;

  for (int i = tid; i < G * HD; i += 256) {
    const int gg = i / HD, d = i % HD;
    qs[gg][d] = bf2f(q[(size_t)b * ldq + (size_t)(g * G + gg) * HD + d]) * scale;
  }

  float m[G], l[G], o[G][8];
#pragma unroll
  for (int gg = 0; gg < G; ++gg) {
    m[gg] = -INFINITY; l[gg] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[gg][j] = 0.f;
  }
  const int dchunk = lane & 15, ksub = lane >> 4;

  for (int pi = s; pi < npages; pi += nsplit) {
    const size_t page = (size_t)block_table[(size_t)b * max_pages + pi];
    const bf16_t* kp = kpool + (page * Hkv + g) * (size_t)(HD / 8) * PAGE * 8;
    const bf16_t* vp = vpool + (page * Hkv + g) * (size_t)PAGE * HD;
    u32x4_t ku[4], vu[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ku[i] = *reinterpret_cast<const u32x4_t*>(kp + ((size_t)(wave * 4 + i) * PAGE + lane) * 8);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int kk = wave * 16 + jj * 4 + ksub;
      vu[jj] = *reinterpret_cast<const u32x4_t*>(vp + (size_t)kk * HD + dchunk * 8);
      if (pi * PAGE + kk >= len) vu[jj] = u32x4_t{0, 0, 0, 0};   // never-written slots: 0 * garbage must stay 0
    }
    __syncthreads();   // qs ready (first page) / sc_part free again (later pages)
    float sc[G];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) sc[gg] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = wave * 4 + i;
      const float kf[8] = {bf_lo(ku[i].x), bf_hi(ku[i].x), bf_lo(ku[i].y), bf_hi(ku[i].y),
                           bf_lo(ku[i].z), bf_hi(ku[i].z), bf_lo(ku[i].w), bf_hi(ku[i].w)};
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const float4 q0 = *reinterpret_cast<const float4*>(&qs[gg][c * 8]);
        const float4 q1 = *reinterpret_cast<const float4*>(&qs[gg][c * 8 + 4]);
        sc[gg] += kf[0] * q0.x + kf[1] * q0.y + kf[2] * q0.z + kf[3] * q0.w + kf[4] * q1.x + kf[5] * q1.y + kf[6] * q1.z + kf[7] * q1.w;
      }
    }
#pragma unroll
    for (int gg = 0; gg < G; ++gg) sc_part[wave][gg][lane] = sc[gg];
    __syncthreads();
    const int key = pi * PAGE + lane;
    float p[G];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      const float full = sc_part[0][gg][lane] + sc_part[1][gg][lane] + sc_part[2][gg][lane] + sc_part[3][gg][lane];
      const float sv = key < len ? full : -INFINITY;
      const float mn = fmaxf(m[gg], wave_max(sv));   // the page has >= 1 valid key, so mn is finite
      const float alpha = __expf(m[gg] - mn);
      p[gg] = __expf(sv - mn);
      l[gg] = l[gg] * alpha + wave_sum(p[gg]);
      m[gg] = mn;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[gg][j] *= alpha;
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int kk = wave * 16 + jj * 4 + ksub;
      const float vf[8] = {bf_lo(vu[jj].x), bf_hi(vu[jj].x), bf_lo(vu[jj].y), bf_hi(vu[jj].y),
                           bf_lo(vu[jj].z), bf_hi(vu[jj].z), bf_lo(vu[jj].w), bf_hi(vu[jj].w)};
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const float pj = __shfl(p[gg], kk, 64);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[gg][j] += pj * vf[j];
      }
    }
  }

  // every wave carries identical (m, l); the o partials (disjoint key subsets) just add up
#pragma unroll
  for (int gg = 0; gg < G; ++gg)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = o[gg][j];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      o[gg][j] = v;
    }
  __syncthreads();
  if (lane < 16) {
#pragma unroll
    for (int gg = 0; gg < G; ++gg)
#pragma unroll
      for (int j = 0; j < 8; ++j) red_o[wave][gg][lane * 8 + j] = o[gg][j];
  }
  __syncthreads();
  for (int i = tid; i < G * HD; i += 256) {
    const int gg = i / HD, d = i % HD;
    const size_t hidx = ((size_t)b * Hq + (g * G + gg)) * nsplit + s;
    part_o[hidx * HD + d] = red_o[0][gg][d] + red_o[1][gg][d] + red_o[2][gg][d] + red_o[3][gg][d];
  }
  if (tid == 0) {
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      const size_t hidx = ((size_t)b * Hq + (g * G + gg)) * nsplit + s;
      part_ml[hidx * 2] = m[gg];
      part_ml[hidx * 2 + 1] = l[gg];
    }
  }
}

// Single-workgroup variant for short/medium contexts: one 512-thread workgroup per (sequence, kv head);
// wave w owns pages w, w + 8, ... (lane = key for Q.K, all 16 K-chunk loads and all 16 V loads of the page in
// flight before any arithmetic), the 8 waves are merged through LDS and the workgroup writes the final bf16
// output - no split partials, no merge pass, so the o_proj GEMV reads a plain 3 KB vector.
template <int G>
__global__ __launch_bounds__(512) void attn_decode_wg_kernel(
    const bf16_t* __restrict__ q, int ldq, const bf16_t* __restrict__ kpool, const bf16_t* __restrict__ vpool,
    const int* __restrict__ block_table, int max_pages, const int* __restrict__ kv_len, int kv_len_add, int Hq, int Hkv,
    float scale, bf16_t* __restrict__ out, int ldo) {
  constexpr int NW = 8;
  __shared__ __attribute__((aligned(16))) float qs[G][HD];
  __shared__ __attribute__((aligned(16))) float red_o[NW][G][HD];
  __shared__ float red_m[NW][G], red_l[NW][G];

  const int b = blockIdx.x / Hkv, g = blockIdx.x % Hkv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int len = kv_len[b] + kv_len_add;
  const int npages = (len + PAGE - 1) / PAGE;
  const int dchunk = lane & 15, ksub = lane >> 4;

  for (int i = tid; i < G * HD; i += 512) {
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

  for (int pi = wave; pi < npages; pi += NW) {
    const size_t page = (size_t)block_table[(size_t)b * max_pages + pi];
    const bf16_t* kp = kpool + (page * Hkv + g) * (size_t)(HD / 8) * PAGE * 8 + (size_t)lane * 8;
    const bf16_t* vp = vpool + (page * Hkv + g) * (size_t)PAGE * HD + (size_t)ksub * HD + dchunk * 8;
    // ---- Q.K: all 16 K-chunk loads of the page in flight, then consumed in order
    float sc[G];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) sc[gg] = 0.f;
    {
      u32x4_t ku[HD / 8];
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) ku[c] = *reinterpret_cast<const u32x4_t*>(kp + (size_t)c * PAGE * 8);
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) {
        const float kf[8] = {bf_lo(ku[c][0]), bf_hi(ku[c][0]), bf_lo(ku[c][1]), bf_hi(ku[c][1]),
                             bf_lo(ku[c][2]), bf_hi(ku[c][2]), bf_lo(ku[c][3]), bf_hi(ku[c][3])};
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
          const float4 q0 = *reinterpret_cast<const float4*>(&qs[gg][c * 8]);
          const float4 q1 = *reinterpret_cast<const float4*>(&qs[gg][c * 8 + 4]);
          sc[gg] += kf[0] * q0.x + kf[1] * q0.y + kf[2] * q0.z + kf[3] * q0.w + kf[4] * q1.x + kf[5] * q1.y + kf[6] * q1.z + kf[7] * q1.w;
        }
        __builtin_amdgcn_sched_barrier(0);   // stop the scheduler hoisting all 192 LDS q reads (it spills otherwise)
      }
    }
    // ---- V rows issued now (the K registers are free): they fly under the softmax shuffles
    u32x4_t vu[PAGE / 4];
#pragma unroll
    for (int jj = 0; jj < PAGE / 4; ++jj) vu[jj] = *reinterpret_cast<const u32x4_t*>(vp + (size_t)jj * 4 * HD);
    __builtin_amdgcn_sched_barrier(0);
    const int key = pi * PAGE + lane;
    float p[G];
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      const float sv = key < len ? sc[gg] : -INFINITY;
      const float mn = fmaxf(m[gg], wave_max(sv));   // the page has >= 1 valid key, so mn is finite
      const float alpha = __expf(m[gg] - mn);
      p[gg] = __expf(sv - mn);
      l[gg] = l[gg] * alpha + wave_sum(p[gg]);
      m[gg] = mn;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[gg][j] *= alpha;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < PAGE / 4; ++jj) {
      const int kk = jj * 4 + ksub;
      u32x4_t vv = vu[jj];
      if (pi * PAGE + kk >= len) vv = u32x4_t{0, 0, 0, 0};   // never-written slots: 0 * garbage must stay 0
      const float vf[8] = {bf_lo(vv[0]), bf_hi(vv[0]), bf_lo(vv[1]), bf_hi(vv[1]),
                           bf_lo(vv[2]), bf_hi(vv[2]), bf_lo(vv[3]), bf_hi(vv[3])};
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const float pj = __shfl(p[gg], kk, 64);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[gg][j] += pj * vf[j];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

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
  for (int i = tid; i < G * HD; i += 512) {
    const int gg = i / HD, d = i % HD;
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) mm = fmaxf(mm, red_m[w][gg]);
    float acc = 0.f, ll = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = red_m[w][gg] == -INFINITY ? 0.f : __expf(red_m[w][gg] - mm);
      acc += f * red_o[w][gg][d];
      ll += f * red_l[w][gg];
    }
    out[(size_t)b * ldo + (size_t)(g * G + gg) * HD + d] = f2bf(acc / ll);
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
  if (!q || !kpool || !vpool || !block_table || !kv_len || !part_o || !part_ml) return VLM_ERR_ARG;
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || nsplit <= 0 || Hq % Hkv != 0) return VLM_ERR_ARG;
  if (D != HD) return VLM_ERR_SHAPE;
  const int G = Hq / Hkv;
  hipStream_t st = (hipStream_t)stream;
  if (nsplit == 1 && out) {
    // one workgroup per (sequence, kv head) writes the final output directly
#define GOW(GV)                                                                                                        \
  hipLaunchKernelGGL((attn_decode_wg_kernel<GV>), dim3(B * Hkv), dim3(512), 0, st, (const bf16_t*)q, ldq,               \
                     (const bf16_t*)kpool, (const bf16_t*)vpool, (const int*)block_table, max_pages, (const int*)kv_len, \
                     kv_len_add, Hq, Hkv, scale, (bf16_t*)out, ldo)
    switch (G) {
      case 1: GOW(1); break;
      case 2: GOW(2); break;
      case 3: GOW(3); break;
      case 4: GOW(4); break;
      case 5: GOW(5); break;
      case 6: GOW(6); break;
      case 7: GOW(7); break;
      default: return VLM_ERR_SHAPE;   // G == 8 spills in this variant: use nsplit > 1
    }
#undef GOW
    VLM_CHECK_LAUNCH();
    return VLM_OK;
  }
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
  if (out) {   // otherwise the caller merges the splits itself (vlm_gemv_attn_out)
    hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(B * Hq), dim3(HD), 0, st, (const float*)part_o,
                       (const float*)part_ml, nsplit, (bf16_t*)out, ldo, Hq);
    VLM_CHECK_LAUNCH();
  }
  return VLM_OK;
}
