// Skinny-M decode GEMM, second form (round 4): the batched decode step's projections (3 <= M <= 16 rows) with the
// ACTIVATIONS IN REGISTERS, two workgroups per CU, and the 4-bit weights loaded coalesced.
//     y[m][n] = epilogue( sum_k x[m][k] * W[n][k] )
// Same call sites as gemv_mfma.hip (the nn.Linear / nn.QuantizedLinear calls of a decoder layer at a batched decode step:
// mlx_vlm/models/qwen2_vl/language.py:52-55,76,120, mlp.py:9-14, utils.py:918-967; RMSNorm language.py:130-133,149-153;
// M-RoPE rope_utils.py:567-651; KVCache.update_and_fetch cache.py:345-367; generate/ar.py:2584-2887), same numerics.
//
// What the first form's counters said (profiles/r03_batch16_sq_pmc_*.txt, r04_phi35v_kv8_kernel_stats.txt): every
// workgroup staged ALL 16 activation rows in LDS (49-115 KB: ONE workgroup = four waves per CU) after normalising them
// itself (two thirds of gate/up's VALU work, repeated by 512 workgroups), then ran a serial chain weights -> MFMA per wave
// with nothing to cover its waits; the 4-bit form read 64-byte pieces of 16 rows per wave instruction (0.5-1.1 TB/s).  Here:
//   * RMSNorm runs ONCE per projection (rmsnorm rows kernel into a scratch of the engine's workspace; the launch it
//     costs is cheaper than 512 repeats of it in front of every weight stream);
//   * a wave owns the K chunks c = c0 + wave + 4 i of its workgroup's K segment for EVERY tile the workgroup walks, so its
//     x^T fragments (lane (m, g): 8 consecutive k of batch row m) are loaded once into registers (16 VGPRs per 128 k) and
//     LDS only holds the per-wave transposition regions and the 4 KB of partial tiles: 22 KB, two workgroups per CU
//     (bounded by 256 VGPRs), eight waves per CU;
//   * a chunk's weights sit in the SAME registers for every tile: the refill for the next tile is issued the moment the
//     chunk has gone to LDS, so a wave always has all its chunks (<= 28 KB) in flight;
//   * bf16: chunk = 16 rows x 128 k, loaded as 4 rows x 256 contiguous bytes per instruction and transposed through the
//     wave's private LDS region (as the first form);
//   * 4-bit (MLX affine, group 64): chunk = 16 rows x 256 k = 128 B of nibbles per row, loaded as 8 rows x 128 contiguous
//     bytes per instruction (full cache lines) and transposed AS PACKED WORDS through the private region (16 bytes per 32
//     weights).  Lane (n, g) then reads the two words at byte 32 G + 8 g of its row for group G: both MFMAs of the pair
//     stay inside one 64-wide group with no cross-lane trade.  A nibble q becomes bf16 128 + q (two per v_and_or_b32), the
//     group enters as scale * (D - 128 sum_x) + bias * sum_x in fp32 - no weight is rounded (the numerics of gemv_w4.hip).
// Units, K segments across workgroups (deterministic last-arriver merge) and epilogues are the first form's.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "internal.h"
#include "../../include/vlm_hip.h"

#ifdef MFMA2_STAMPS
// measurement build only (scripts/mfma2_probe.hip): wall-clock stamps (100 MHz) of wave 0 of workgroup MFMA2_STAMPS
__device__ unsigned long long g_mfma2_stamps[64];
#define ST2(i) do { if (blockIdx.x == MFMA2_STAMPS && threadIdx.x == 0 && (i) < 64) g_mfma2_stamps[i] = wall_clock64(); } while (0)
#else
#define ST2(i)
#endif

namespace {

constexpr int MEPI2_ROPE_KV = 1 << 10;
constexpr int WREG2 = 16 * 272;      // bytes of a wave's private transposition region (4-bit: 16 x 144 + 256 of scales)

struct Mfma2Args {
  const bf16_t *x, *W, *bias, *res, *norm_w;
  float eps;
  const unsigned* Wsb;   // 4-bit: W = q words uint32 [N][K/8], Wsb = (scale | bias << 16) [N][K/64]; else null
  bf16_t* y;
  int M, N, K, ldx, ldw, ldy, ldres;
  VlmRopeKv rk;
  float* ws;             // [n_tiles * KS][256] fp32 partial tiles (KS > 1)
  unsigned* tickets;     // [n_tiles] arrivals of the current launch (zero between launches)
  int n_tiles, KS, nchunk, cps;   // row tiles, K segments, chunks of K, chunks per segment
};

template <int EPI>
__device__ __forceinline__ int tile_row2(const Mfma2Args& a, int tile, int r) {
  if (EPI == MEPI2_ROPE_KV) {
    const int half = a.rk.D >> 1, tph = half >> 3, n_rot = (a.rk.Hq + a.rk.Hkv) * tph;
    if (tile < n_rot) return (tile / tph) * a.rk.D + (tile % tph) * 8 + (r & 7) + (r >> 3) * half;
    return (a.rk.Hq + a.rk.Hkv) * a.rk.D + (tile - n_rot) * 16 + r;
  }
  return tile * 16 + r;
}

// NCH: chunks per wave and unit (the wave's chunk slots).  bf16: chunk = 128 k (4 MFMA steps, 4 weight loads, 16 x VGPRs);
// 4-bit: chunk = 256 k (8 MFMA steps, 2 weight loads + 1 scale word, 32 x VGPRs).
// NW: waves per workgroup (4 or 8; 8 halves a wave's x^T fragments: the 4-bit form's 32 VGPRs per chunk).
// NSETS: register sets of weights, i.e. units of the workgroup in flight per wave (a set is refilled for the unit NSETS
// ahead the moment its chunk has gone to LDS).  With ONE set a wave's next unit is requested when the current one
// ARRIVES: a unit per memory round trip - the 4-bit form (2.25 KB per chunk) then has 27 KB per CU in flight and runs at
// 1.2-1.7 TB/s (profiles/r04_mfma2_shapes.txt); its sets cost 9 VGPRs per chunk, so it takes four.
// NORM: the RMSNorm prologue inside the launch (one K segment only: the workgroup's waves hold ALL of x): every wave takes
// the sum of squares of its fragments, the partials meet in LDS, the fragments are normalised in registers - behind the
// weight loads, which are already in flight.  (With K segments the rows are normalised by one rows kernel in front.)
template <int EPI, bool W4, int NCH, int NW, int NSETS, bool NORM>
__global__ __launch_bounds__(64 * NW, 2) void gemv_mfma2_kernel(const Mfma2Args a) {
  constexpr int NS = W4 ? 8 : 4;       // MFMA k steps per chunk
  constexpr int NJ = W4 ? 2 : 4;       // weight load instructions per chunk
  constexpr int CK = W4 ? 256 : 128;   // k per chunk
  __shared__ __attribute__((aligned(16))) char s_wreg[NW * WREG2];
  __shared__ float part[NW * 256];
  __shared__ int flags[64];
  __shared__ float red[NORM ? NW * 16 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  char* wreg = s_wreg + wave * WREG2;
  unsigned* sbw = reinterpret_cast<unsigned*>(wreg + 16 * 144);      // 4-bit: [4 groups][16 rows] (scale | bias) words of a chunk
  const int n_units = a.n_tiles * a.KS, G = gridDim.x;
  const int ks = (int)blockIdx.x % a.KS;                                // G % KS == 0: a workgroup keeps its K segment
  const int c0 = ks * a.cps, c1 = min(a.nchunk, c0 + a.cps);
  const int n_valid = max(0, min(NCH, (c1 - c0 - wave + NW - 1) / NW));     // this wave's chunks c0 + wave + NW i < c1
  const int mrow = min(r16, a.M - 1);
  ST2(0);

  // ---- x^T fragments of the wave's chunks (rows past M alias row M - 1: their columns of D are dropped)
  u32x4_t xf[NCH][NS];
  {
    const bf16_t* xr = a.x + (size_t)mrow * a.ldx;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = min(c0 + wave + NW * i, c1 - 1);                   // surplus slots re-read the last chunk, never multiplied
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        // bf16: step s of the chunk = k 32 s + 8 g; 4-bit: step (G, w) = (s >> 1, s & 1): k 64 G + 16 g + 8 w
        const int k = c * CK + (W4 ? 64 * (s >> 1) + 16 * g + 8 * (s & 1) : 32 * s + 8 * g);
        xf[i][s] = *reinterpret_cast<const u32x4_t*>(xr + k);
      }
    }
  }

  // weights of (unit, slot): the wave's chunk of the unit's 16 rows, coalesced
  u32x4_t wv[NSETS][NCH][NJ];
  unsigned sbv[NSETS][NCH];
  auto load_slot = [&](int u, int i, u32x4_t (&wv)[NCH][NJ], unsigned (&sbv)[NCH]) __attribute__((always_inline)) {
    const int tile = u / a.KS;
    const int c = min(c0 + wave + NW * i, c1 - 1);
    if (W4) {
      // instruction j: rows 8 j + (lane >> 3), bytes 16 (lane & 7) of the row's 128-byte chunk
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const size_t row = (size_t)min(tile_row2<EPI>(a, tile, 8 * j + (lane >> 3)), a.N - 1);
        const unsigned* wq = reinterpret_cast<const unsigned*>(a.W) + row * (a.K >> 3) + (size_t)c * 32 + (lane & 7) * 4;
        wv[i][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wq));
      }
      const size_t srow = (size_t)min(tile_row2<EPI>(a, tile, lane >> 2), a.N - 1);
      sbv[i] = a.Wsb[srow * (a.K >> 6) + (size_t)c * 4 + (lane & 3)];
    } else {
      // instruction j: rows 4 j + g, bytes 16 r16 of the row's 256-byte chunk
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bf16_t* wr = a.W + (size_t)min(tile_row2<EPI>(a, tile, 4 * j + g), a.N - 1) * a.ldw + (size_t)c * 128 + r16 * 8;
        wv[i][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wr));
      }
    }
  };

  int u = blockIdx.x;
  __builtin_amdgcn_sched_barrier(0);      // the activation loads first (vector loads return in issue order), the stream behind
  // (the further sets go out behind the prologue when it runs here: its temporaries and four sets in flight do not fit)
  constexpr int PRE = NORM ? 1 : NSETS;
#pragma unroll
  for (int s_ = 0; s_ < PRE; ++s_)
    if (u + s_ * G < n_units) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) load_slot(u + s_ * G, i, wv[s_], sbv[s_]);
    }
  __builtin_amdgcn_sched_barrier(0);

  if (NORM) {
    // nn.RMSNorm's typed graph on the fragments: bf16(x * inv) * weight -> bf16 (language.py:130-133); inv from the fp32 sum
    // of squares of the whole row (this wave's chunks, the four k groups of a row in lanes r16 + 16 g, then the NW waves in
    // a fixed order)
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (i < n_valid) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const u32x4_t v = xf[i][s];
#pragma unroll
          for (int q = 0; q < 4; ++q) ss += bf_lo(v[q]) * bf_lo(v[q]) + bf_hi(v[q]) * bf_hi(v[q]);
        }
      }
    ss = col4_sum(ss);
    if (g == 0) red[wave * 16 + r16] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[w * 16 + r16];
    const float inv = rsqrtf(tot / (float)a.K + a.eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = min(c0 + wave + NW * i, c1 - 1);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int k = c * CK + (W4 ? 64 * (s >> 1) + 16 * g + 8 * (s & 1) : 32 * s + 8 * g);
        const u32x4_t wu = *reinterpret_cast<const u32x4_t*>(a.norm_w + k);
        const u32x4_t v = xf[i][s];
        u32x4_t o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = pack_bf2(bf_lo(wu[q]) * rbf(bf_lo(v[q]) * inv), bf_hi(wu[q]) * rbf(bf_hi(v[q]) * inv));
        xf[i][s] = o;
      }
    }
  }

  // 4-bit: group sums of the wave's activations (per batch row m = r16 and 64-wide group) and the expanded-word order
  float sx[W4 ? NCH : 1][4];
  if (W4) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
#pragma unroll
      for (int Gq = 0; Gq < 4; ++Gq) {
        float p = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const u32x4_t v = xf[i][W4 ? 2 * Gq + w : 0];
          p += ((bf_lo(v[0]) + bf_hi(v[0])) + (bf_lo(v[1]) + bf_hi(v[1]))) + ((bf_lo(v[2]) + bf_hi(v[2])) + (bf_lo(v[3]) + bf_hi(v[3])));
        }
        sx[W4 ? i : 0][Gq] = col4_sum(p);          // the four 16-lane rows hold the four quarters of the group
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        // 8 consecutive bf16 -> the order (0,4,1,5,2,6,3,7) of an expanded q word
        const u32x4_t v = xf[i][s];
        u32x4_t o;
        o[0] = (v[0] & 0xffffu) | (v[2] << 16);
        o[1] = (v[0] >> 16) | (v[2] & 0xffff0000u);
        o[2] = (v[1] & 0xffffu) | (v[3] << 16);
        o[3] = (v[1] >> 16) | (v[3] & 0xffff0000u);
        xf[i][s] = o;
      }
    }
  }

  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s_ = PRE; s_ < NSETS; ++s_)
    if (u + s_ * G < n_units) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) load_slot(u + s_ * G, i, wv[s_], sbv[s_]);
    }
  __builtin_amdgcn_sched_barrier(0);

  // epilogue of one finished 16 x 16 tile: thread tid holds element (n_l = tid >> 4, m = tid & 15); whole workgroup
  auto finish = [&](int tile, float v) __attribute__((always_inline)) {
    const int n_l = (tid >> 4) & 15, m = tid & 15;
    const int n = tile_row2<EPI>(a, tile, n_l);
    if (EPI == MEPI2_ROPE_KV || (EPI & VLM_EPI_SWIGLU)) {
      __syncthreads();                  // every thread has read its part[] sums
      if (tid < 256) part[tid] = v;
      __syncthreads();
    }
    if (NW > 4 && tid >= 256) return;   // (after the barriers: threads 0..255 hold the tile)
    if (EPI == MEPI2_ROPE_KV) {
      const int half = a.rk.D >> 1, tph = half >> 3, n_rot = (a.rk.Hq + a.rk.Hkv) * tph;
      // SuScaledRoPE's per-call rule (rope_utils.py:168-172): long factors for every row of the step once ANY row's cache
      // offset has reached original_max
      const float* inv_tab = a.rk.inv_freq;
      if (a.rk.long_from > 0) {
        bool any_long = false;
        for (int mm = 0; mm < a.M; ++mm) any_long |= a.rk.slot[mm] >= a.rk.long_from;
        inv_tab += any_long ? half : 0;
      }
      if (m < a.M) {
        const int e_slot = a.rk.slot[m];
        const size_t e_page = a.rk.block_table ? (size_t)a.rk.block_table[(size_t)m * a.rk.max_pages + (e_slot >> 6)]
                                               : (size_t)m * a.rk.max_pages + (e_slot >> 6);
        const int e_within = e_slot & 63;
        if (tile < n_rot) {
          if (n_l < 8) {
            const int head = tile / tph, j = (tile % tph) * 8 + n_l, n1 = n + half;
            // (4-bit: quantized_matmul rounds to bf16, the bias add is a second typed op)
            const float y0 = rbf((W4 ? rbf(v) : v) + bf2f(a.bias[n])), y1 = rbf((W4 ? rbf(part[tid + 128]) : part[tid + 128]) + bf2f(a.bias[n1]));
            float sn, cs;
            sincosf((float)a.rk.pos[m] * inv_tab[j], &sn, &cs);
            const float z0 = rbf(y0 * a.rk.qk_scale), z1 = rbf(y1 * a.rk.qk_scale);      // SuScaledRoPE's typed x * scale
            const float o0 = z0 * cs - z1 * sn, o1 = z1 * cs + z0 * sn;
            if (head < a.rk.Hq) {
              a.y[(size_t)m * a.ldy + n] = f2bf(o0);
              a.y[(size_t)m * a.ldy + n1] = f2bf(o1);
            } else {
              const int gq = head - a.rk.Hq, d0 = j, d1 = j + half;
              bf16_t* kb = a.rk.kpool + (e_page * a.rk.Hkv + gq) * (size_t)(a.rk.D >> 3) * 512;
              kb[((size_t)(d0 >> 3) * 64 + e_within) * 8 + (d0 & 7)] = f2bf(o0);
              kb[((size_t)(d1 >> 3) * 64 + e_within) * 8 + (d1 & 7)] = f2bf(o1);
            }
          }
        } else if (n < a.N) {
          const int vr = n - (a.rk.Hq + a.rk.Hkv) * a.rk.D, gq = vr / a.rk.D, d = vr % a.rk.D;
          bf16_t* vb = a.rk.vpool + ((e_page * a.rk.Hkv + gq) * (size_t)a.rk.D + d) * 64 + vlm_vslot(e_within);
          vb[0] = f2bf(rbf((W4 ? rbf(v) : v) + bf2f(a.bias[n])));
        }
      }
    } else if (EPI & VLM_EPI_SWIGLU) {
      if (m < a.M && !(n_l & 1) && n + 1 < a.N)
        a.y[(size_t)m * a.ldy + (n >> 1)] = f2bf(swiglu_(rbf(v), rbf(part[tid + 16])));
    } else if (m < a.M && n < a.N) {
      if (EPI & VLM_EPI_BIAS) v = (W4 ? rbf(v) : v) + bf2f(a.bias[n]);
      if (EPI & VLM_EPI_RESIDUAL) v = rbf(v) + bf2f(a.res[(size_t)m * a.ldres + n]);
      a.y[(size_t)m * a.ldy + n] = f2bf(v);
    }
  };

  ST2(1);
  [[maybe_unused]] int st_k = 0;     // (phase-stamp index of the -DMFMA2_STAMPS probe build)
  auto unit = [&](int u, u32x4_t (&wv)[NCH][NJ], unsigned (&sbv)[NCH]) __attribute__((always_inline)) {
    const int un = u + NSETS * G;         // the unit this set is refilled for
    const bool more = un < n_units;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    // chunk slot i: registers -> the wave's private region (same-wave LDS operations execute in order: no barrier), the
    // refill for the next unit goes out at once, then fragments and MFMAs
    auto chunk = [&](int i) __attribute__((always_inline)) {
      if (W4) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) *reinterpret_cast<u32x4_t*>(wreg + (8 * j + (lane >> 3)) * 144 + (lane & 7) * 16) = wv[i][j];
        sbw[(lane & 3) * 16 + (lane >> 2)] = sbv[W4 ? i : 0];
        if (more) load_slot(un, i, wv, sbv);
#pragma unroll
        for (int Gq = 0; Gq < 4; ++Gq) {
          const uint2 wd = *reinterpret_cast<const uint2*>(wreg + r16 * 144 + Gq * 32 + g * 8);
          f32x4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const unsigned q = w ? wd.y : wd.x;
            const u32x4_t af = {(q & 0x000F000Fu) | 0x43004300u, ((q >> 4) & 0x000F000Fu) | 0x43004300u,
                                ((q >> 8) & 0x000F000Fu) | 0x43004300u, ((q >> 12) & 0x000F000Fu) | 0x43004300u};
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, xf[i][W4 ? 2 * Gq + w : 0]), d,
                                                        0, 0, 0);
          }
          const u32x4_t s4 = *reinterpret_cast<const u32x4_t*>(sbw + Gq * 16 + 4 * g);     // rows 4 g .. 4 g + 3 of group Gq
          const float sxg = sx[W4 ? i : 0][Gq];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] += bf_lo(s4[q]) * (d[q] - 128.f * sxg) + bf_hi(s4[q]) * sxg;
        }
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) *reinterpret_cast<u32x4_t*>(wreg + (4 * j + g) * 272 + r16 * 16) = wv[i][j];
        if (more) load_slot(un, i, wv, sbv);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const u32x4_t af = *reinterpret_cast<const u32x4_t*>(wreg + r16 * 272 + kb * 64 + g * 16);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, xf[i][W4 ? 0 : kb]), acc,
                                                        0, 0, 0);
        }
      }
    };
    if (n_valid == NCH) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) chunk(i);
    } else {
#pragma unroll
      for (int i = 0; i < NCH; ++i)
        if (i < n_valid) chunk(i);                 // (wave-uniform)
    }

    // D[n = 4 g + q][m = r16]  ->  part[wave][n * 16 + m]
#pragma unroll
    for (int q = 0; q < 4; ++q) part[wave * 256 + (4 * g + q) * 16 + r16] = acc[q];
    ST2(2 + 3 * st_k);
    __syncthreads();
    ST2(3 + 3 * st_k);
    const int te = tid & 255;
    float v = (part[te] + part[256 + te]) + (part[512 + te] + part[768 + te]);
    if (NW == 8) v += (part[1024 + te] + part[1280 + te]) + (part[1536 + te] + part[1792 + te]);
    if (a.KS > 1) {
      // partial tile -> workspace, agent scope (write-through); tickets and merges after the workgroup's LAST unit
      if (tid < 256) __hip_atomic_store(a.ws + (size_t)u * 256 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      finish(u, v);
    }
    __syncthreads();                      // part[] is rewritten by the next unit
    ST2(4 + 3 * st_k);
    ++st_k;
  };
  while (u < n_units) {
#pragma unroll
    for (int s_ = 0; s_ < NSETS; ++s_) {
      if (u < n_units) unit(u, wv[s_], sbv[s_]);      // (uniform)
      u += G;
    }
  }

  ST2(62);
  if (a.KS > 1) {
    // deferred hand-off (as gemv_mfma.hip): ONE wait, the tickets of all units of this workgroup at once (thread i: unit
    // blockIdx.x + i G), then the merges of the tiles it arrived last at - partials summed in the fixed order
    // ks = 0 .. KS - 1, so the result does not depend on who merges
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int n_mine = ((int)blockIdx.x < n_units) ? (n_units - (int)blockIdx.x + G - 1) / G : 0;      // <= 64 (host)
    if (tid < n_mine) {
      const int tile_i = ((int)blockIdx.x + tid * G) / a.KS;
      const unsigned t = __hip_atomic_fetch_add(a.tickets + tile_i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flags[tid] = t == (unsigned)a.KS - 1u;
    }
    __syncthreads();
    for (int i = 0; i < n_mine; ++i) {
      if (!flags[i]) continue;            // (uniform)
      const int tile_i = ((int)blockIdx.x + i * G) / a.KS;
      float v = 0.f;
      for (int k = 0; k < a.KS; ++k)
        v += __hip_atomic_load(a.ws + ((size_t)tile_i * a.KS + k) * 256 + (tid & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid == 0) __hip_atomic_store(a.tickets + tile_i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      finish(tile_i, v);
      __syncthreads();
    }
  }
}

template <int EPI, bool W4, int NCH, int NW, int NSETS, bool NORM>
int launch2n(const Mfma2Args& a, int n_units, hipStream_t st) {
  auto kern = gemv_mfma2_kernel<EPI, W4, NCH, NW, NSETS, NORM>;
  static int nb = 0;
  if (nb == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64 * NW, 0) != hipSuccess || nb < 1)) nb = 1;
  static const int wgs_per_cu = [] { const char* e = getenv("VLM_GEMV_MFMA2_WGS_PER_CU"); return e ? max(1, min(8, atoi(e))) : 3; }();
  int grid = min(n_units, 256 * min(nb, wgs_per_cu));
  grid -= grid % a.KS;
  if (grid <= 0) return -1;
  if ((n_units + grid - 1) / grid > 64) return -1;      // tickets of a workgroup's units: one thread each, 64 flags
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

template <int EPI, bool W4, int NCH, int NW, int NSETS>
int launch2(const Mfma2Args& a, int n_units, hipStream_t st) {
  constexpr bool CAN_NORM = (EPI & VLM_EPI_RESIDUAL) == 0;
  if constexpr (CAN_NORM) {
    if (a.norm_w) return launch2n<EPI, W4, NCH, NW, NSETS, true>(a, n_units, st);
  }
  return launch2n<EPI, W4, NCH, NW, NSETS, false>(a, n_units, st);
}


template <int EPI, bool W4>
int launch2_nch(const Mfma2Args& a, int nch, int n_units, hipStream_t st) {
  if constexpr (W4) {
    switch (nch) {
      case 1: return launch2<EPI, true, 1, 8, 4>(a, n_units, st);
      case 2: return launch2<EPI, true, 2, 8, 4>(a, n_units, st);
      default: return -1;
    }
  } else {
    switch (nch) {
      case 1: return launch2<EPI, false, 1, 4, 2>(a, n_units, st);
      case 2: return launch2<EPI, false, 2, 4, 2>(a, n_units, st);
      case 3: return launch2<EPI, false, 3, 4, 1>(a, n_units, st);
      case 4: return launch2<EPI, false, 4, 4, 1>(a, n_units, st);
      case 5: return launch2<EPI, false, 5, 4, 1>(a, n_units, st);
      case 6: return launch2<EPI, false, 6, 4, 1>(a, n_units, st);
      default: return -1;
    }
  }
}

}  // namespace

#ifdef VLM_MFMA2_W4_TU
#define VLM_MFMA2_ENTRY vlm_gemv_mfma2_try_w4
constexpr bool kW4 = true;
#else
#define VLM_MFMA2_ENTRY vlm_gemv_mfma2_try_bf16
constexpr bool kW4 = false;
#endif

// -> VLM_OK, an error, or -1: shape not handled here (the caller takes the first form).  norm_w != nullptr: the RMSNorm
// prologue, run here as ONE rows kernel into the workspace's activation scratch (needs ws and contiguous rows).
VLM_INTERNAL int VLM_MFMA2_ENTRY(const void* x, const void* W, const void* Wsb, const void* bias, const void* res, const void* norm_w,
                                 void* y, int M, int N, int K, int ldx, int ldw, int ldy, int ldres, float eps, int epilogue,
                                 const VlmRopeKv* rk, void* ws, void* stream) {
  constexpr int CK = kW4 ? 256 : 128;
  const int NW = kW4 ? 8 : 4, NCH_MAX = kW4 ? 2 : (rk ? 5 : 6);      // what fits 256 VGPRs without spills (x^T fragments + the chunks in flight)
  if ((Wsb != nullptr) != kW4) return -1;
  if (M < 1 || M > 16 || K % CK || ldx % 8 || (!kW4 && ldw % 8)) return -1;
  const bool rope = rk != nullptr;
  if (rope && (rk->D % 16 || !bias || !norm_w)) return -1;
  if (norm_w && (!ws || ldx != K || K > 8192 || (epilogue & VLM_EPI_RESIDUAL))) return -1;      // (ws: the rows kernel's scratch)
  if (!rope && epilogue != VLM_EPI_NONE && epilogue != VLM_EPI_BIAS && epilogue != VLM_EPI_RESIDUAL && epilogue != VLM_EPI_SWIGLU &&
      epilogue != (VLM_EPI_BIAS | VLM_EPI_RESIDUAL))
    return -1;
  if ((epilogue & VLM_EPI_SWIGLU) && (N % 16)) return -1;
  Mfma2Args a{};
  a.x = (const bf16_t*)x; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias; a.res = (const bf16_t*)res;
  a.y = (bf16_t*)y; a.Wsb = (const unsigned*)Wsb;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldres = ldres;
  if (rope) {
    a.rk = *rk;
    if (N != (rk->Hq + 2 * rk->Hkv) * rk->D) return -1;
    a.n_tiles = (rk->Hq + rk->Hkv) * (rk->D / 16) + rk->Hkv * rk->D / 16;
  } else {
    a.n_tiles = vlm_cdiv(N, 16);
  }
  a.nchunk = K / CK;
  // K segments: as few as keep a wave's chunks within its registers (NW x NCH_MAX chunks per workgroup and segment); few row
  // tiles split further while a segment still gives every wave two chunks (units >= ~2 per CU keep the chip streaming)
  static const int ks_env = [] { const char* e = getenv("VLM_GEMV_MFMA2_KS"); return e ? atoi(e) : 0; }();
  int KS = vlm_cdiv(a.nchunk, NW * NCH_MAX);
  static const int min_units = [] { const char* e = getenv("VLM_GEMV_MFMA2_MIN_UNITS"); return e ? atoi(e) : 384; }();
  while (ws && a.n_tiles * KS < min_units && vlm_cdiv(a.nchunk, KS + 1) >= 8 && KS < 16) ++KS;
  if (ks_env > 0) KS = max(KS, min(ks_env, a.nchunk));
  if (KS > 1 && (!ws || (size_t)a.n_tiles * KS > 4096 || a.n_tiles > 8192)) return -1;
  a.KS = KS;
  a.cps = vlm_cdiv(a.nchunk, KS);
  KS = a.KS = vlm_cdiv(a.nchunk, a.cps);            // no empty segment
  const int nch = vlm_cdiv(a.cps, NW);
  if (nch > NCH_MAX) return -1;
  // Which projections take this form (measured per projection at 16 rows, profiles/r04_mfma2_shapes.txt; VLM_GEMV_MFMA2=2
  // takes it wherever it is legal): every 4-bit one (Phi-3.5 layer 74.4 -> 53.6 us); bf16 without a norm prologue unless
  // the registers force a K split the first form does not need (7B o_proj: 28 chunks, 9.6 vs 11.9 us); bf16 with a norm
  // prologue from K = 4096 (Mistral gate/up 67.8 -> 54.4 us; at 2B / 7B widths the first form's in-launch prologue wins:
  // 5.9 vs 6.2, 14.3 vs 16.3, 16.4 vs 18.8 us)
  static const int force = [] { const char* e = getenv("VLM_GEMV_MFMA2"); return e ? atoi(e) : 1; }();
  if (!kW4 && force != 2) {
    if (norm_w && K < 4096) return -1;
    if (!norm_w && a.nchunk <= 28 && vlm_cdiv(a.nchunk, 4) > NCH_MAX) return -1;
  }
  a.ws = (float*)ws;
  a.tickets = ws ? (unsigned*)((char*)ws + (size_t)4096 * 256 * 4) : nullptr;
  const int n_units = a.n_tiles * KS;
  static const bool debug = [] { const char* e = getenv("VLM_GEMV_MFMA_DEBUG"); return e && atoi(e) != 0; }();
  if (debug)
    fprintf(stderr, "[gemv_mfma2] M=%d N=%d K=%d %s epi=%d: tiles=%d KS=%d cps=%d nch=%d units=%d\n", M, N, K, kW4 ? "w4" : "bf16",
            rope ? -1 : epilogue, a.n_tiles, KS, a.cps, nch, n_units);
  hipStream_t st = (hipStream_t)stream;
  // measured (profiles/r04_mfma2_probe_v1.txt): in the launch 6.2 vs 7.4 us (2B qkv), 16.3 vs 16.5 (2B gate/up) - but the 4-bit
  // form's 8-wave workgroups spend 8.9 us in it (19.9 -> 21.2 us at Phi-3.5 gate/up): rows kernel there
  static const int norm_env = [] { const char* e = getenv("VLM_GEMV_MFMA2_NORM_IN_KERNEL"); return e ? atoi(e) : -1; }();
  const bool norm_in_kernel = norm_env >= 0 ? norm_env != 0 : !kW4;
  if (norm_w && KS == 1 && norm_in_kernel) {
    a.norm_w = (const bf16_t*)norm_w;      // one K segment: the workgroup holds whole rows, the prologue runs in the launch
    a.eps = eps;
  } else if (norm_w) {
    // nn.RMSNorm's typed graph (bf16(x * inv) * weight -> bf16), once for all workgroups: rows [M][K] after the tickets
    bf16_t* xn = reinterpret_cast<bf16_t*>((char*)ws + VLM_MFMA_WS_XN_OFFSET);
    const int rc = vlm_rmsnorm_residual(x, nullptr, norm_w, xn, nullptr, M, K, eps, stream);
    if (rc != VLM_OK) return rc;
    a.x = xn;
    a.ldx = K;
  }
  if (rope) return launch2_nch<MEPI2_ROPE_KV, kW4>(a, nch, n_units, st);
  switch (epilogue) {
    case VLM_EPI_NONE: return launch2_nch<VLM_EPI_NONE, kW4>(a, nch, n_units, st);
    case VLM_EPI_BIAS: return launch2_nch<VLM_EPI_BIAS, kW4>(a, nch, n_units, st);
    case VLM_EPI_RESIDUAL: return launch2_nch<VLM_EPI_RESIDUAL, kW4>(a, nch, n_units, st);
    case VLM_EPI_BIAS | VLM_EPI_RESIDUAL: return launch2_nch<VLM_EPI_BIAS | VLM_EPI_RESIDUAL, kW4>(a, nch, n_units, st);
    case VLM_EPI_SWIGLU: return launch2_nch<VLM_EPI_SWIGLU, kW4>(a, nch, n_units, st);
    default: return -1;
  }
}
