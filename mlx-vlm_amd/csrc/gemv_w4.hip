// Decode-time GEMV over MLX affine 4-bit weights for gfx950 (HBM-bound, dequantisation fused into the operand path):
//     y[m][n] = epilogue( sum_k prologue(x)[m][k] * (scale[n][k/64] * q[n][k] + bias[n][k/64]) ),   m < MB <= 8
//
// Replaces nn.QuantizedLinear.__call__ = mx.quantized_matmul(x, weight, scales, biases, transpose=True, group_size=64,
// bits=4) (+ bias) - what the reference's load_model turns every Linear of a 4-bit checkpoint into
// (mlx_vlm/utils.py:918-967, nn.quantize with the `<path>.scales in weights` predicate) - at L == 1, with the same
// fusions as the bf16 GEMV family (csrc/gemv_bf16.hip): RMSNorm prologue (language.py:130-133,149-153,200), bias,
// residual (151-153), SwiGLU on interleaved gate/up rows (mlp.py:6-14), M-RoPE + paged KV write (rope_utils.py:567-651,
// cache.py:345-367).
//
// Layout (repacked once at load, models/quantized.py): q words uint32 [N][K/8] exactly as MLX stores them (element k of a
// row in word k / 8, bits 4 (k % 8) .. +3); scale and bias of a 64-wide group in ONE uint32 [N][K/64] (scale bf16 in the
// low half, bias bf16 in the high half), so a lane's 32-element chunk costs one 16-byte and one 4-byte load.
//
// Arithmetic.  A nibble q in [0, 15] is turned into the bf16 number 128 + q by OR-ing it into the mantissa of 0x4300
// (128.0: ulp 1), two per instruction: (w >> 4 j) & 0x000F000F | 0x43004300 holds elements j and j + 4 of a word.  With x
// stored in LDS in the matching pair order, v_dot2c_f32_bf16 accumulates sum (128 + q_k) x_k in fp32; per 32-element
// chunk   scale * (dot - 128 * sum x) + bias * sum x   is the group's exact affine form (fp32; sum x is computed once per
// lane and chunk, shared by all rows).  No weight is ever rounded to bf16.
#include "common.hpp"
#include "internal.h"
#include "../../include/vlm_hip.h"

namespace {

enum { WPRO_NONE = 0, WPRO_RMSNORM = 1 };
constexpr int WEPI_ROPE_KV = 1 << 10;

struct W4RopeKv {
  const int* pos;
  const int* slot;
  const float* inv_freq;
  const int* block_table;
  int max_pages, Hq, Hkv, D;
  bf16_t* kpool;
  bf16_t* vpool;
  float qk_scale = 1.f;      // SuScaledRoPE: q / k times this (typed op) before the rotation
  int long_from = 0;         // > 0: inv_freq = [2][D/2] (short, long); long for the whole step when any row's slot >= long_from
};

__device__ __forceinline__ float wdot2(unsigned w, unsigned x, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, x), acc, false);
}

// one q word (8 nibbles) against the 8 x values of its block, stored as pairs (x_j, x_{j+4}) in xp[0..3]
__device__ __forceinline__ float w4_word(unsigned w, const u32x4_t xp, float acc) {
  acc = wdot2((w & 0x000F000Fu) | 0x43004300u, xp[0], acc);
  acc = wdot2(((w >> 4) & 0x000F000Fu) | 0x43004300u, xp[1], acc);
  acc = wdot2(((w >> 8) & 0x000F000Fu) | 0x43004300u, xp[2], acc);
  acc = wdot2(((w >> 12) & 0x000F000Fu) | 0x43004300u, xp[3], acc);
  return acc;
}

// 8 consecutive bf16 (natural order, 16 bytes) -> the pair order of w4_word
__device__ __forceinline__ uint4 pair_order(uint4 v) {
  uint4 o;
  o.x = (v.x & 0xffffu) | (v.z << 16);            // (x0, x4)
  o.y = (v.x >> 16) | (v.z & 0xffff0000u);        // (x1, x5)
  o.z = (v.y & 0xffffu) | (v.w << 16);            // (x2, x6)
  o.w = (v.y >> 16) | (v.w & 0xffff0000u);        // (x3, x7)
  return o;
}

// lane chunk = 32 elements (4 q words); KC chunks per lane cover K <= 2048 * KC; R rows per wave and pass.
// A workgroup stages the activation rows ONCE and then walks row groups gw = blockIdx.x * 4 + wave, + 4 * gridDim.x, ...
// (the launcher caps the grid at 8 workgroups per CU): with one group per workgroup every one of the 1120 (gate/up) or
// 9496 (head) workgroups repeated the x load + RMSNorm + LDS round trip before its first FMA - 8.7 us for 7.7 MB and
// 3.4 TB/s on the head (profiles/r02_w4_kernel_stats.txt).  The next group's weights are in flight (second register
// set, when R * KC <= 4) while the current one is reduced.
template <int R, int KC, int MB, int PRO, int EPI>
__global__ __launch_bounds__(256) void gemv_w4_kernel(const bf16_t* __restrict__ x, const unsigned* __restrict__ Wq,
                                                      const unsigned* __restrict__ Wsb, const bf16_t* __restrict__ bias,
                                                      const bf16_t* __restrict__ res, const bf16_t* __restrict__ norm_w,
                                                      bf16_t* __restrict__ y, int N, int K, int ldx, int ldy, int ldres,
                                                      float eps, W4RopeKv rk, int n_groups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // xs[MB][K] bf16 in pair order | red[8] f32
  constexpr bool DBUF = R * KC <= 4;        // long rows (KC >= 5) are the few-row projections: one pass, no second set
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nch32 = K >> 5, nch8 = K >> 3, ngrp = K >> 6;
  const int gstride = gridDim.x * 4;

  // ---- rows of a group (same maps as gemv_bf16.hip)
  auto rows_of = [&](int gw, int (&row)[R]) {
    if (EPI == WEPI_ROPE_KV) {
      const int half = rk.D >> 1, n_pair = (rk.Hq + rk.Hkv) * half;
      if (gw < n_pair) {
        row[0] = (gw / half) * rk.D + gw % half;
        row[R - 1] = row[0] + half;
      } else {
        row[0] = (rk.Hq + rk.Hkv) * rk.D + 2 * (gw - n_pair);
        row[R - 1] = row[0] + 1;
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) row[r] = gw * R + r;
    }
  };
  // ---- every q chunk and its (scale, bias) word of a group's rows
  auto load_w = [&](const int (&row)[R], u32x4_t (&wq)[R][KC], unsigned (&sb)[R][KC]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const size_t rr = (size_t)min(row[r], N - 1);
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int ch = min(lane + 64 * c, nch32 - 1);
        wq[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(Wq + rr * nch8 + (size_t)ch * 4));
        sb[r][c] = Wsb[rr * ngrp + (ch >> 1)];
      }
    }
  };

  int gw = blockIdx.x * 4 + wave;
  int row[R];
  u32x4_t wq[R][KC];
  unsigned sb[R][KC];
  rows_of(min(gw, n_groups - 1), row);
  load_w(row, wq, sb);                                   // the weight stream first
  __builtin_amdgcn_sched_barrier(0);

  // ---- activation rows -> LDS as bf16 in pair order (normalised when PRO_RMSNORM): x and the norm weight are read ONCE
  //      (one global round trip), the sum of squares comes from the registers
  float* red = reinterpret_cast<float*>(smem + (size_t)MB * K * 2);
  constexpr int XCH = KC >= 2 ? 2 : 1;       // norm prologue: K <= 4096 (checked by the launcher); KC == 1 <=> K <= 2048
  if (PRO == WPRO_RMSNORM) {
    uint4 xv[MB][XCH], wv[XCH];
#pragma unroll
    for (int u = 0; u < XCH; ++u) {
      const int i = min(tid + 256 * u, nch8 - 1);
      wv[u] = reinterpret_cast<const uint4*>(norm_w)[i];
#pragma unroll
      for (int m = 0; m < MB; ++m) xv[m][u] = reinterpret_cast<const uint4*>(x + (size_t)m * ldx)[i];
    }
    float ssq[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < XCH; ++u)
        if (tid + 256 * u < nch8) {
          const uint4 v = xv[m][u];
          const float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
#pragma unroll
          for (int j = 0; j < 8; ++j) s += f[j] * f[j];
        }
      ssq[m] = wave_sum(s);
    }
    if (lane == 0)
#pragma unroll
      for (int m = 0; m < MB; ++m) red[wave * MB + m] = ssq[m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const float inv = rsqrtf((red[m] + red[MB + m] + red[2 * MB + m] + red[3 * MB + m]) / (float)K + eps);
      uint4* xs = reinterpret_cast<uint4*>(smem + (size_t)m * K * 2);
#pragma unroll
      for (int u = 0; u < XCH; ++u)
        if (tid + 256 * u < nch8) {
          uint4 v = xv[m][u];
          const uint4 wu = wv[u];
          // nn.RMSNorm typed graph: bf16(x * inv) then * weight -> bf16 (as gemv_bf16.hip)
          v.x = pack_bf2(bf_lo(wu.x) * rbf(bf_lo(v.x) * inv), bf_hi(wu.x) * rbf(bf_hi(v.x) * inv));
          v.y = pack_bf2(bf_lo(wu.y) * rbf(bf_lo(v.y) * inv), bf_hi(wu.y) * rbf(bf_hi(v.y) * inv));
          v.z = pack_bf2(bf_lo(wu.z) * rbf(bf_lo(v.z) * inv), bf_hi(wu.z) * rbf(bf_hi(v.z) * inv));
          v.w = pack_bf2(bf_lo(wu.w) * rbf(bf_lo(v.w) * inv), bf_hi(wu.w) * rbf(bf_hi(v.w) * inv));
          xs[tid + 256 * u] = pair_order(v);
        }
    }
  } else {
    for (int m = 0; m < MB; ++m) {
      const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)m * ldx);
      uint4* xs = reinterpret_cast<uint4*>(smem + (size_t)m * K * 2);
      for (int i = tid; i < nch8; i += 256) xs[i] = pair_order(xr[i]);
    }
  }
  __syncthreads();

  for (; gw < n_groups; gw += gstride) {
    // the next group's weights: in flight while this one is multiplied and reduced
    int row_n[R];
    u32x4_t wq_n[DBUF ? R : 1][DBUF ? KC : 1];
    unsigned sb_n[DBUF ? R : 1][DBUF ? KC : 1];
    const bool more = gw + gstride < n_groups;
    if constexpr (DBUF) {
      rows_of(min(gw + gstride, n_groups - 1), row_n);
      if (more) load_w(row_n, reinterpret_cast<u32x4_t(&)[R][KC]>(wq_n), reinterpret_cast<unsigned(&)[R][KC]>(sb_n));
      __builtin_amdgcn_sched_barrier(0);
    }

    float acc[R][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch32) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          const uint4* xs = reinterpret_cast<const uint4*>(smem + (size_t)m * K * 2) + (size_t)ch * 4;
          // (ext_vector registers, constant indices only: plain arrays passed by pointer end up in scratch)
          const u32x4_t* xv = reinterpret_cast<const u32x4_t*>(xs);
          const u32x4_t xp0 = xv[0], xp1 = xv[1], xp2 = xv[2], xp3 = xv[3];
          float sx = 0.f;
#define SX4(V) ((bf_lo(V[0]) + bf_hi(V[0])) + (bf_lo(V[1]) + bf_hi(V[1])) + (bf_lo(V[2]) + bf_hi(V[2])) + (bf_lo(V[3]) + bf_hi(V[3])))
          sx = SX4(xp0) + SX4(xp1) + SX4(xp2) + SX4(xp3);
#undef SX4
#pragma unroll
          for (int r = 0; r < R; ++r) {
            float d = 0.f;
            const u32x4_t wv = wq[r][c];
            d = w4_word(wv[0], xp0, d);
            d = w4_word(wv[1], xp1, d);
            d = w4_word(wv[2], xp2, d);
            d = w4_word(wv[3], xp3, d);
            const float sc = bf_lo(sb[r][c]), bi = bf_hi(sb[r][c]);
            acc[r][m] += sc * (d - 128.f * sx) + bi * sx;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[r][m] = wave_sum(acc[r][m]);

    if (EPI == WEPI_ROPE_KV) {
      // R == 2: (d, d + D/2) of one q / k head, or two consecutive v rows; lane m stores batch row m (as gemv_bf16.hip)
      const int half = rk.D >> 1, n_pair = (rk.Hq + rk.Hkv) * half;
      const bool rope_pair = gw < n_pair;
      const int rope_head = gw / half, rope_j = gw % half;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        a0 = (lane == m) ? acc[0][m] : a0;
        a1 = (lane == m) ? acc[R - 1][m] : a1;
      }
      // SuScaledRoPE's per-call rule (rope_utils.py:168-172): the long factors for every row once ANY row's cache offset
      // has reached original_max (wave-uniform vote over the rows of the step)
      const bool use_long = rk.long_from > 0 && __any(rk.slot[min(lane, MB - 1)] >= rk.long_from);
      const float* inv_tab = rk.inv_freq + (use_long ? (rk.D >> 1) : 0);
      if (lane < MB) {
        const int m = lane;
        const int r0 = min(row[0], N - 1), r1 = min(row[R - 1], N - 1);
        // quantized_matmul rounds to bf16, the bias add is a second typed op
        const float y0 = rbf(rbf(a0) + bf2f(bias[r0])), y1 = rbf(rbf(a1) + bf2f(bias[r1]));
        const int e_slot = rk.slot[m], e_pos = rk.pos[m];
        const size_t e_page = rk.block_table ? (size_t)rk.block_table[(size_t)m * rk.max_pages + (e_slot >> 6)]
                                             : (size_t)m * rk.max_pages + (e_slot >> 6);
        const int e_within = e_slot & 63;
        if (rope_pair) {
          float sn, cs;
          sincosf((float)e_pos * inv_tab[rope_j], &sn, &cs);
          const float z0 = rbf(y0 * rk.qk_scale), z1 = rbf(y1 * rk.qk_scale);
          const float o0 = z0 * cs - z1 * sn, o1 = z1 * cs + z0 * sn;
          if (rope_head < rk.Hq) {
            y[(size_t)m * ldy + row[0]] = f2bf(o0);
            y[(size_t)m * ldy + row[R - 1]] = f2bf(o1);
          } else {
            const int g = rope_head - rk.Hq, d0 = rope_j, d1 = rope_j + (rk.D >> 1);
            bf16_t* kb = rk.kpool + (e_page * rk.Hkv + g) * (size_t)(rk.D >> 3) * 512;
            kb[((size_t)(d0 >> 3) * 64 + e_within) * 8 + (d0 & 7)] = f2bf(o0);
            kb[((size_t)(d1 >> 3) * 64 + e_within) * 8 + (d1 & 7)] = f2bf(o1);
          }
        } else {
          const int vr = row[0] - (rk.Hq + rk.Hkv) * rk.D, g = vr / rk.D, d = vr % rk.D;
          bf16_t* vb = rk.vpool + ((e_page * rk.Hkv + g) * (size_t)rk.D + d) * 64 + vlm_vslot(e_within);
          vb[0] = f2bf(y0);
          vb[64] = f2bf(y1);
        }
      }
    } else if (EPI & VLM_EPI_SWIGLU) {
#pragma unroll
      for (int r = 0; r < R; r += 2)
#pragma unroll
        for (int m = 0; m < MB; ++m)
          if (lane == (r >> 1) * MB + m && row[r] + 1 < N)
            y[(size_t)m * ldy + (row[r] >> 1)] = f2bf(swiglu_(rbf(acc[r][m]), rbf(acc[r + 1][m])));
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m)
          if (lane == r * MB + m && row[r] < N) {
            float v = acc[r][m];
            if (EPI & VLM_EPI_BIAS) v = rbf(v) + bf2f(bias[row[r]]);          // quantized_matmul rounds, then + bias (typed op)
            if (EPI & VLM_EPI_RESIDUAL) v = rbf(v) + bf2f(res[(size_t)m * ldres + row[r]]);
            y[(size_t)m * ldy + row[r]] = f2bf(v);
          }
    }

    if constexpr (DBUF) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        row[r] = row_n[r];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          wq[r][c] = wq_n[r][c];
          sb[r][c] = sb_n[r][c];
        }
      }
    } else if (more) {
      rows_of(gw + gstride, row);
      load_w(row, wq, sb);
    }
  }
}

// W4 -> bf16 rows (mx.dequantize: scale * q + bias, fp32 multiply then add - not contracted - one rounding).  rows ==
// nullptr: rows 0..n_rows-1 of the matrix (prefill: the weight is materialised once per GEMM into a scratch matrix);
// rows != nullptr: gather (nn.QuantizedEmbedding: dequantize of the looked-up rows).  One thread = one q word (8 weights).
__global__ __launch_bounds__(256) void dequant_w4_kernel(const unsigned* __restrict__ Wq, const unsigned* __restrict__ Wsb,
                                                         const int* __restrict__ rows, bf16_t* __restrict__ out, int n_rows,
                                                         int K, int ldo, int n_table_rows) {
  const int nch8 = K >> 3;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_rows * nch8) return;
  const int r = (int)(idx / nch8), c = (int)(idx % nch8);
  int src = rows ? rows[r] : r;
  src = src < 0 ? 0 : (src >= n_table_rows ? n_table_rows - 1 : src);
  const unsigned w = Wq[(size_t)src * nch8 + c], sbw = Wsb[(size_t)src * (K >> 6) + (c >> 3)];
  const float sc = bf_lo(sbw), bi = bf_hi(sbw);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = __fadd_rn(__fmul_rn(sc, (float)((w >> (4 * j)) & 0xFu)), bi);
  uint4 o;
  o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
  *reinterpret_cast<uint4*>(out + (size_t)r * ldo + (size_t)c * 8) = o;
}

struct W4Args {
  const void *x, *Wq, *Wsb, *bias, *res, *norm_w;
  void* y;
  int N, K, ldx, ldy, ldres;
  float eps;
  W4RopeKv rk;
  hipStream_t st;
};

inline int w4_err() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

template <int R, int KC, int MB, int PRO, int EPI>
int w4_launch(const W4Args& a, int n_waves) {
  const size_t lds = (size_t)MB * a.K * 2 + 256;      // + red[4][MB] f32
  auto kern = gemv_w4_kernel<R, KC, MB, PRO, EPI>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return VLM_ERR_HIP + (int)e;
  }
  static int per_cu = 0;                                    // resident workgroups per CU of this instantiation
  if (per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 2;
    per_cu = nb > 8 ? 8 : nb;
  }
  const int grid = min(vlm_cdiv(n_waves, 4), 256 * per_cu);   // the resident workgroups walk the row groups
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, a.st, (const bf16_t*)a.x, (const unsigned*)a.Wq,
                     (const unsigned*)a.Wsb, (const bf16_t*)a.bias, (const bf16_t*)a.res, (const bf16_t*)a.norm_w,
                     (bf16_t*)a.y, a.N, a.K, a.ldx, a.ldy, a.ldres, a.eps, a.rk, n_waves);
  return w4_err();
}

template <int KC, int MB, int PRO, int EPI>
int w4_rows(const W4Args& a) {
  if (EPI == WEPI_ROPE_KV) {
    const int waves = (a.rk.Hq + a.rk.Hkv) * (a.rk.D / 2) + a.rk.Hkv * a.rk.D / 2;
    return w4_launch<2, KC, MB, PRO, EPI>(a, waves);
  }
  // 4 rows per wave once there are enough rows to give every CU several workgroups; KC * R * 5 registers of weights
  if (a.N >= 8192 && KC <= 2) return w4_launch<4, KC, MB, PRO, EPI>(a, vlm_cdiv(a.N, 4));
  return w4_launch<2, KC, MB, PRO, EPI>(a, vlm_cdiv(a.N, 2));
}

template <int MB, int PRO, int EPI>
int w4_k(const W4Args& a) {
  if (a.K <= 2048) return w4_rows<1, MB, PRO, EPI>(a);
  if (a.K <= 4096) return w4_rows<2, MB, PRO, EPI>(a);
  if (a.K <= 10240) return w4_rows<5, MB, PRO, EPI>(a);
  if (a.K <= 20480) return w4_rows<10, MB, PRO, EPI>(a);
  return VLM_ERR_SHAPE;
}

template <int PRO, int EPI>
int w4_m(int M, const W4Args& a) {
  switch (M) {
    case 1: return w4_k<1, PRO, EPI>(a);
    case 2: return w4_k<2, PRO, EPI>(a);
    case 4: return w4_k<4, PRO, EPI>(a);
    case 8: return w4_k<8, PRO, EPI>(a);
    default: return VLM_ERR_SHAPE;
  }
}

}  // namespace

extern "C" int vlm_gemv_w4(const void* x, const void* Wq, const void* Wsb, const void* bias, const void* res,
                           const void* norm_w, void* y, int M, int N, int K, int ldx, int ldy, int ldres, float eps,
                           int epilogue, void* stream) {
  return vlm_gemv_w4_ex(x, Wq, Wsb, bias, res, norm_w, y, M, N, K, ldx, ldy, ldres, eps, epilogue, 1, nullptr, stream);
}

extern "C" int vlm_gemv_w4_ws(const void* x, const void* Wq, const void* Wsb, const void* bias, const void* res, const void* norm_w,
                              void* y, int M, int N, int K, int ldx, int ldy, int ldres, float eps, int epilogue, void* workspace,
                              void* stream) {
  return vlm_gemv_w4_ex(x, Wq, Wsb, bias, res, norm_w, y, M, N, K, ldx, ldy, ldres, eps, epilogue, 1, workspace, stream);
}

extern "C" int vlm_gemv_w4_qkv_rope_kvwrite_ws(const void* h, const void* norm_w, float eps, const void* Wq, const void* Wsb,
                                               const void* bqkv, void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D,
                                               const void* pos, const void* slot, const void* inv_freq, const void* block_table,
                                               int max_pages, void* kpool, void* vpool, void* workspace, void* stream) {
  return vlm_gemv_w4_qkv_rope_kvwrite_ex(h, norm_w, eps, Wq, Wsb, bqkv, qkv, ldq, M, hidden, Hq, Hkv, D, pos, slot, inv_freq,
                                         block_table, max_pages, kpool, vpool, 1, workspace, 1.f, 0, stream);
}

// mfma / ws as vlm_gemv_bf16_ex: batched steps (5..16 rows) go to the dequant-fused MFMA form of csrc/gemv_mfma.hip
VLM_INTERNAL int vlm_gemv_w4_ex(const void* x, const void* Wq, const void* Wsb, const void* bias, const void* res,
                                const void* norm_w, void* y, int M, int N, int K, int ldx, int ldy, int ldres, float eps,
                                int epilogue, int mfma, void* ws, void* stream) {
  if (!x || !Wq || !Wsb || !y || N <= 0 || K <= 0) return VLM_ERR_ARG;
  if (mfma) {
    const int rc = vlm_gemv_mfma_try_w4(x, Wq, Wsb, bias, res, norm_w, y, M, N, K, ldx, ldy, ldres, eps, epilogue, nullptr, ws, stream);
    if (rc >= 0) return rc;
  }
  if ((epilogue & VLM_EPI_BIAS) && !bias) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_RESIDUAL) && !res) return VLM_ERR_ARG;
  if (K % 64 != 0 || ldx % 8 != 0) return VLM_ERR_SHAPE;
  if ((epilogue & VLM_EPI_SWIGLU) && (N % 2 != 0)) return VLM_ERR_SHAPE;
  if ((size_t)M * K * 2 + 256 > 160 * 1024) return VLM_ERR_SHAPE;
  if (norm_w && K > 4096) return VLM_ERR_SHAPE;      // the norm prologue keeps x in registers: 2 chunks per thread
  W4Args a{x, Wq, Wsb, bias, res, norm_w, y, N, K, ldx, ldy, ldres, eps, W4RopeKv{}, (hipStream_t)stream};
#define GO(P, E) return w4_m<P, E>(M, a)
  if (norm_w) {
    switch (epilogue) {
      case VLM_EPI_NONE: GO(WPRO_RMSNORM, VLM_EPI_NONE);
      case VLM_EPI_BIAS: GO(WPRO_RMSNORM, VLM_EPI_BIAS);
      case VLM_EPI_SWIGLU: GO(WPRO_RMSNORM, VLM_EPI_SWIGLU);
      default: return VLM_ERR_ARG;
    }
  }
  switch (epilogue) {
    case VLM_EPI_NONE: GO(WPRO_NONE, VLM_EPI_NONE);
    case VLM_EPI_BIAS: GO(WPRO_NONE, VLM_EPI_BIAS);
    case VLM_EPI_RESIDUAL: GO(WPRO_NONE, VLM_EPI_RESIDUAL);
    case VLM_EPI_SWIGLU: GO(WPRO_NONE, VLM_EPI_SWIGLU);
    default: return VLM_ERR_ARG;
  }
#undef GO
}

extern "C" int vlm_gemv_w4_qkv_rope_kvwrite(const void* h, const void* norm_w, float eps, const void* Wq, const void* Wsb,
                                            const void* bqkv, void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D,
                                            const void* pos, const void* slot, const void* inv_freq,
                                            const void* block_table, int max_pages, void* kpool, void* vpool,
                                            void* stream) {
  return vlm_gemv_w4_qkv_rope_kvwrite_ex(h, norm_w, eps, Wq, Wsb, bqkv, qkv, ldq, M, hidden, Hq, Hkv, D, pos, slot, inv_freq,
                                         block_table, max_pages, kpool, vpool, 1, nullptr, 1.f, 0, stream);
}

VLM_INTERNAL int vlm_gemv_w4_qkv_rope_kvwrite_ex(const void* h, const void* norm_w, float eps, const void* Wq, const void* Wsb,
                                                 const void* bqkv, void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D,
                                                 const void* pos, const void* slot, const void* inv_freq,
                                                 const void* block_table, int max_pages, void* kpool, void* vpool, int mfma,
                                                 void* ws, float qk_scale, int long_from, void* stream) {
  if (!h || !norm_w || !Wq || !Wsb || !bqkv || !qkv || !pos || !slot || !inv_freq || !kpool || !vpool || max_pages <= 0)
    return VLM_ERR_ARG;
  if (mfma) {
    const VlmRopeKv rk{(const int*)pos, (const int*)slot, (const float*)inv_freq, (const int*)block_table, max_pages, Hq, Hkv, D,
                       (unsigned short*)kpool, (unsigned short*)vpool, qk_scale, long_from};
    const int rc = vlm_gemv_mfma_try_w4(h, Wq, Wsb, bqkv, nullptr, norm_w, qkv, M, (Hq + 2 * Hkv) * D, hidden, hidden, ldq, 0, eps,
                                        VLM_EPI_BIAS, &rk, ws, stream);
    if (rc >= 0) return rc;
  }
  if (hidden % 64 || hidden > 4096 || D % 16 || (size_t)M * hidden * 2 > 64 * 1024) return VLM_ERR_SHAPE;
  const int N = (Hq + 2 * Hkv) * D;
  W4Args a{h, Wq, Wsb, bqkv, nullptr, norm_w, qkv, N, hidden, hidden, ldq, 0, eps,
           W4RopeKv{(const int*)pos, (const int*)slot, (const float*)inv_freq, (const int*)block_table, max_pages, Hq, Hkv, D,
                    (bf16_t*)kpool, (bf16_t*)vpool, qk_scale, long_from},
           (hipStream_t)stream};
  return w4_m<WPRO_RMSNORM, WEPI_ROPE_KV>(M, a);
}

extern "C" int vlm_dequant_w4(const void* Wq, const void* Wsb, const void* rows, void* out, int n_rows, int K, int ldo,
                              int n_table_rows, void* stream) {
  if (!Wq || !Wsb || !out || n_rows < 0 || K <= 0 || n_table_rows <= 0) return VLM_ERR_ARG;
  if (K % 64 != 0 || ldo % 8 != 0) return VLM_ERR_SHAPE;
  if (n_rows == 0) return VLM_OK;
  const long total = (long)n_rows * (K / 8);
  hipLaunchKernelGGL(dequant_w4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned*)Wq, (const unsigned*)Wsb, (const int*)rows, (bf16_t*)out, n_rows, K, ldo, n_table_rows);
  return w4_err();
}
