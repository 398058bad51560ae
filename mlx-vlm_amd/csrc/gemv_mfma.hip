// Skinny-M decode GEMM for gfx950: the weight-streaming GEMV family of gemv_bf16.hip for 3 <= M <= 16 batch rows
//     y[m][n] = epilogue( sum_k prologue(x)[m][k] * W[n][k] )
// on the matrix cores.  Same call sites (the nn.Linear calls of a decoder layer at L = 1, mlx_vlm/models/qwen2_vl/
// language.py:52-55,76,120, mlp.py:9-14, language.py:514-517 as_linear; RMSNorm 130-133,149-153,200; M-RoPE rope_utils.py:
// 567-651; KVCache.update_and_fetch cache.py:345-367), used by a batched decode step (generate/ar.py:2584-2887).
//
// Why: the v_dot2c kernels re-read every activation chunk from LDS and issue M dot products per weight chunk and row: at
// M = 8 the VALU, not HBM, sets their pace (gate/up 22 us vs 10.6 us at M = 1).  Here the batch rows are the N dimension
// of v_mfma_f32_16x16x32_bf16: one instruction multiplies a 16-row x 32-k weight fragment with all (<= 16) batch rows, so
// the kernel stays a weight stream for every M it accepts.
//
// Mapping.  A = W tile (16 output rows x 32 k), B = x^T (32 k x 16 batch rows; rows past M alias row M - 1 and their
// columns of D are dropped), D[n][m] fp32.  An A fragment wants lane l to hold row l & 15 - ADJACENT lanes on DIFFERENT rows,
// i.e. 16-byte pieces 3 KB apart when loaded straight from the weight matrix: the first version did that and ran 2.5 x
// SLOWER than the v_dot2c kernels (27 us for the 4.7 MB o_proj: every wave instruction is 64 uncoalesced accesses,
// profiles/r02_mfma_gemv.txt).  So the weights are loaded COALESCED (one instruction = 4 rows x 256 contiguous bytes),
// all of a wave's chunks in flight before the first use, and each wave transposes chunk by chunk through a PRIVATE 4 KiB
// LDS region (row pitch 272 B: conflict-free ds_read_b128 fragments; same-wave LDS operations execute in order, so the
// region needs no barrier).  x^T fragments come from the workgroup's staged activations.
//   unit   = (16-row tile, K segment ks of KS); the 4 waves of a workgroup interleave the unit's 128-wide K chunks (<= 7
//            each, up to 28 loads of a lane in flight), their partial tiles meet in LDS
//   KS > 1 (few row tiles x long K: o_proj, down): fp32 partial tiles go to a workspace, the LAST workgroup of a tile
//            to arrive (agent-scope ticket) sums them in the fixed order ks = 0..KS-1 - deterministic - and runs the epilogue;
//            the tickets and merges of a workgroup come after its LAST unit (round 3: the first version ended EVERY unit with
//            partial store -> vmcnt(0) -> ticket -> partial loads), and which projections split at all follows the measured
//            policy in mfma_try (profiles/r03_mfma_shapes.txt)
//   the activations (normalised when the RMSNorm prologue is on) are staged ONCE per workgroup, which then walks units
//            blockIdx.x, + gridDim.x, ...: all of K (FULLX), or - long K, no norm: the down projection - only the K
//            segment ks = blockIdx.x % KS that all its units share (the grid is a multiple of KS)
// Epilogues as gemv_bf16.hip, one thread per (n, m) of the tile: bias, residual, SwiGLU on interleaved gate / up rows, and
// M-RoPE + paged KV write, for which a tile's 16 rows are (d0 .. d0 + 7, d0 + D/2 .. d0 + D/2 + 7) of one q / k head so
// that both elements of a rotation pair sit in the tile.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "internal.h"
#include "../../include/vlm_hip.h"

namespace {

enum { MPRO_NONE = 0, MPRO_RMSNORM = 1 };
constexpr int MEPI_ROPE_KV = 1 << 10;
constexpr int NCW = 7;             // 128-wide K chunks per wave and unit (28 per unit: K = 3584, the 7B hidden size, in one segment)
constexpr int WREG = 16 * 272;      // bytes of a wave's private transposition region

struct MfmaArgs {
  const bf16_t *x, *W, *bias, *res, *norm_w;
  const unsigned* Wsb;  // MLX affine 4-bit weights: W = q words uint32 [N][K/8], Wsb = (scale | bias << 16) [N][K/64]; else null
  bf16_t* y;
  int M, N, K, ldx, ldw, ldy, ldres;
  float eps;
  VlmRopeKv rk;
  float* ws;            // [n_tiles * KS][256] fp32 partial tiles (KS > 1)
  unsigned* tickets;    // [n_tiles] arrivals of the current launch (zero between launches)
  int n_tiles, KS, nblk, bpk;   // row tiles, K segments, 128-wide chunks of K, chunks per segment
  int y_tiled;                  // 1: y in the tiled layout [N_out / 8][16][8] the row-slice form reads (gemv_mfma_rows.hip; VLM_EPI_Y_TILED)
};

template <int EPI>
__device__ __forceinline__ int tile_row(const MfmaArgs& a, int tile, int r) {
  if (EPI == MEPI_ROPE_KV) {
    const int half = a.rk.D >> 1, tph = half >> 3, n_rot = (a.rk.Hq + a.rk.Hkv) * tph;
    if (tile < n_rot) return (tile / tph) * a.rk.D + (tile % tph) * 8 + (r & 7) + (r >> 3) * half;
    return (a.rk.Hq + a.rk.Hkv) * a.rk.D + (tile - n_rot) * 16 + r;
  }
  return tile * 16 + r;
}

// NCW: 128-wide K chunks per wave and unit (3: three register sets, the next two units' weights are in flight while this
// one is multiplied; 7: one set).  XS > 0: the activations fit the prologue's registers (rows_per_wave * chunks_per_lane <=
// XS): x and the norm weight are loaded ONCE, ahead of the weight stream (vector loads return in issue order), and the RMS
// statistics come from the registers; XS < 0 (round 3): the row-trip form of the same for any row count - a wave loads 2 or
// 4 of its rows whole per trip (K <= 4096 / <= 1536 staged elements per row); XS == 0: the loop form (longer rows only: one
// dependent L2 round trip per chunk and row).
//
// W4 (MLX affine 4-bit weights, group 64 - nn.QuantizedLinear at a batched decode step, reference utils.py:918-967): a
// chunk is 64 bytes of nibbles per row, ONE 16-byte load per lane + the row's (scale | bias) pair.  A q word IS 8
// consecutive k of one row, i.e. an A fragment of the MFMA: lane (row l & 15, quarter l >> 4) loads the 4 words of its
// row's quarter - the weights go from global memory to the matrix cores through registers only, no LDS transposition (the
// bf16 form needs one because a 16-byte piece of a bf16 row is only a quarter of a fragment's k range).  The contraction
// index is free, so the 4 MFMAs of a chunk take one word each; for the affine form every MFMA must stay inside one 64-wide
// group, while a lane's 16 bytes lie in one group only: the lower and upper 32 lanes trade two words (lanes l and l ^ 32,
// one cross-lane move per pair of words), after which words 0 / 1 of every lane belong to group 0 and words 2 / 3 to group
// 1; the x^T fragments are read at the matching k offsets.  A nibble q becomes the bf16 number 128 + q by OR-ing it into the
// mantissa of 0x4300, two per instruction, which leaves the 8 weights of a word in the order (0,4,1,5,2,6,3,7) - the
// activations are staged in the same order.  Each group is multiplied on its own (2 MFMAs from a zero accumulator) and
// enters the sum as scale * (D - 128 * sum_x) + bias * sum_x in fp32 (sum_x per batch row and group, once per workgroup):
// the exact affine form, no weight is rounded - the numerics of csrc/gemv_w4.hip.  A wave's chunks are independent
// straight-line code when the unit is full (no per-chunk branch): MFMAs, dequantisation and LDS reads of neighbouring
// chunks overlap.
template <int PRO, int EPI, bool FULLX, int NCW, int XS, bool W4>
__global__ __launch_bounds__(256) void gemv_mfma_kernel(const MfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // xs[M][P] | wreg[4][WREG] | part[4][256] f32 | red | flag | sbr | xsum
  // register sets of weights in flight: a unit's loads go out NSETS - 1 units before it is multiplied
  constexpr int NSETS = NCW == 3 ? 3 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  const int kx = FULLX ? a.K : a.bpk * 128;                // staged elements per batch row
  const int P = kx * 2 + 16;                               // row pitch (bytes): P / 16 odd -> conflict-free ds_read_b128
  char* wreg = smem + (size_t)a.M * P + (size_t)wave * WREG;
  float* part = reinterpret_cast<float*>(smem + (size_t)a.M * P + 4 * WREG);
  float* red = part + 1024;
  int* s_flag = reinterpret_cast<int*>(red + 64);
  unsigned* sbr = reinterpret_cast<unsigned*>(red + 80) + wave * (7 * 32);   // W4: per chunk slot, the wave's 16 rows x 2 groups of (scale | bias)
  float* xsum = red + 80 + 4 * 7 * 32;                                      // W4: [M][kx / 64] sums of the staged activations
  const int n_units = a.n_tiles * a.KS, G = gridDim.x;
  const int mrow = min(r16, a.M - 1);

  // chunk c of the unit (128 k): instruction j covers rows 4 j .. 4 j + 3, lane -> row 4 j + (lane >> 4), 16 bytes at
  // k offset 8 (lane & 15): 256 contiguous bytes per row.  W4: one instruction covers all 16 rows (lane -> row lane >> 2,
  // 16 bytes = 32 nibbles at k offset 32 (lane & 3)); slot [1] carries the row's (scale | bias) word of group (lane & 1)
  auto load_w = [&](int u, u32x4_t (&wv)[NCW][4]) {
    const int tile = u / a.KS, ks = u % a.KS;
    const int kb0 = ks * a.bpk, kb1 = min(a.nblk, kb0 + a.bpk);
    if (W4) {
      // lane (r16, g): row r16 of the tile, the 4 words (32 k) of quarter g of the chunk; slot [1] = the row's (scale | bias)
      // words of the chunk's two groups
      const size_t row = (size_t)min(tile_row<EPI>(a, tile, r16), a.N - 1);
      const unsigned* wq = reinterpret_cast<const unsigned*>(a.W) + row * (a.K >> 3) + g * 4;
      const unsigned* sb = a.Wsb + row * (a.K >> 6);
#pragma unroll
      for (int i = 0; i < NCW; ++i) {
        const int b = max(min(kb0 + wave + 4 * i, kb1 - 1), 0);
        wv[i][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wq + (size_t)b * 16));
        const uint2 s2 = *reinterpret_cast<const uint2*>(sb + (size_t)b * 2);
        wv[i][1][0] = s2.x;
        wv[i][1][1] = s2.y;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16_t* wr = a.W + (size_t)min(tile_row<EPI>(a, tile, 4 * j + g), a.N - 1) * a.ldw + r16 * 8;
#pragma unroll
      for (int i = 0; i < NCW; ++i) {
        const int b = max(min(kb0 + wave + 4 * i, kb1 - 1), 0);   // clamped: surplus slots re-read the last chunk, unused
        wv[i][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wr + (size_t)b * 128));
      }
    }
  };

  // W4: 8 consecutive bf16 -> the order (0,4,1,5,2,6,3,7) of an expanded q word
  auto x_order = [&](u32x4_t v) {
    if (!W4) return v;
    u32x4_t o;
    o[0] = (v[0] & 0xffffu) | (v[2] << 16);
    o[1] = (v[0] >> 16) | (v[2] & 0xffff0000u);
    o[2] = (v[1] & 0xffffu) | (v[3] << 16);
    o[3] = (v[1] >> 16) | (v[3] & 0xffff0000u);
    return o;
  };
  // W4: per batch row and 64-wide group, the sum of the staged (bf16) activations; after the staging barrier
  auto group_sums = [&]() {
    if (!W4) return;
    const int ng = kx >> 6;
    for (int i = tid; i < a.M * ng; i += 256) {
      const int m = i / ng, q = i % ng;
      const u32x4_t* p = reinterpret_cast<const u32x4_t*>(smem + (size_t)m * P + (size_t)q * 128);
      float sx = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4_t v = p[c];
        sx += ((bf_lo(v[0]) + bf_hi(v[0])) + (bf_lo(v[1]) + bf_hi(v[1]))) + ((bf_lo(v[2]) + bf_hi(v[2])) + (bf_lo(v[3]) + bf_hi(v[3])));
      }
      xsum[i] = sx;
    }
    __syncthreads();
  };

  // ---- activations -> LDS (bf16, rows at pitch P), loop form: before any weight load when it is the FULLX prologue
  auto stage_x = [&](int kb0) {
    const int nch = kx >> 3, k0 = FULLX ? 0 : kb0 * 128;
    if (PRO == MPRO_RMSNORM) {
      for (int m = wave; m < a.M; m += 4) {                  // sum of squares per batch row: wave w takes rows w, w + 4, ...
        const uint4* xr = reinterpret_cast<const uint4*>(a.x + (size_t)m * a.ldx);
        float ss = 0.f;
        for (int i = lane; i < nch; i += 64) {
          const uint4 v = xr[i];
          const float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
#pragma unroll
          for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
        }
        ss = wave_sum(ss);
        if (lane == 0) red[m] = rsqrtf(ss / (float)a.K + a.eps);
      }
      __syncthreads();
    }
    for (int i = tid; i < a.M * nch; i += 256) {
      const int m = i / nch, c = i % nch;
      const int kc = min(k0 + c * 8, a.K - 8);               // segment tail past K: re-read, never multiplied
      uint4 v = *reinterpret_cast<const uint4*>(a.x + (size_t)m * a.ldx + kc);
      if (PRO == MPRO_RMSNORM) {
        const float inv = red[m];
        const uint4 wu = *reinterpret_cast<const uint4*>(a.norm_w + kc);
        // nn.RMSNorm typed graph: bf16(x * inv) then * weight -> bf16 (as gemv_bf16.hip)
        v.x = pack_bf2(bf_lo(wu.x) * rbf(bf_lo(v.x) * inv), bf_hi(wu.x) * rbf(bf_hi(v.x) * inv));
        v.y = pack_bf2(bf_lo(wu.y) * rbf(bf_lo(v.y) * inv), bf_hi(wu.y) * rbf(bf_hi(v.y) * inv));
        v.z = pack_bf2(bf_lo(wu.z) * rbf(bf_lo(v.z) * inv), bf_hi(wu.z) * rbf(bf_hi(v.z) * inv));
        v.w = pack_bf2(bf_lo(wu.w) * rbf(bf_lo(v.w) * inv), bf_hi(wu.w) * rbf(bf_hi(v.w) * inv));
      }
      const u32x4_t o = x_order(u32x4_t{v.x, v.y, v.z, v.w});
      *reinterpret_cast<u32x4_t*>(smem + (size_t)m * P + (size_t)c * 16) = o;
    }
  };

  int u = blockIdx.x;
  u32x4_t wv[NSETS][NCW][4];
  auto& wvA = wv[0];
  const int kb_fix = FULLX ? 0 : (blockIdx.x % a.KS) * a.bpk;      // !FULLX: this workgroup's K segment (G % KS == 0)
  if (XS > 0) {
    // wave w owns batch rows w, w + 4, ..; slot s = (row index rw, chunk column cl): chunk lane + 64 cl of row w + 4 rw
    constexpr int XA = XS > 0 ? XS : 1, NWC = XS > 6 ? 7 : 3;   // chunks per lane <= 3 (K <= 1536) in the 6-slot form
    const int nch = kx >> 3, cpl = (nch + 63) >> 6, k0 = kb_fix * 128;
    u32x4_t xr[XA], nw[NWC];     // (ext_vector arrays: HIP's uint4 struct arrays end up in scratch)
#pragma unroll
    for (int s_ = 0; s_ < XA; ++s_) {
      const int rw = s_ / cpl, cl = s_ % cpl;
      const int m = min(wave + 4 * rw, a.M - 1), c = min(lane + 64 * cl, nch - 1);
      xr[s_] = *reinterpret_cast<const u32x4_t*>(a.x + (size_t)m * a.ldx + min(k0 + c * 8, a.K - 8));
    }
    if (PRO == MPRO_RMSNORM) {
#pragma unroll
      for (int cl = 0; cl < NWC; ++cl) nw[cl] = reinterpret_cast<const u32x4_t*>(a.norm_w)[min(lane + 64 * min(cl, cpl - 1), nch - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
    load_w(min(u, n_units - 1), wvA);                        // behind the small loads: they return first
    __builtin_amdgcn_sched_barrier(0);
    const int rpw = (a.M + 3) >> 2;
    for (int rw = 0; rw < rpw; ++rw) {
      const int m = wave + 4 * rw;
      float inv = 1.f;
      if (PRO == MPRO_RMSNORM) {
        float ss = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < XA; ++s_)
          if (s_ / cpl == rw && lane + 64 * (s_ % cpl) < nch) {
            const u32x4_t v = xr[s_];
            const float f[8] = {bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1]), bf_lo(v[2]), bf_hi(v[2]), bf_lo(v[3]), bf_hi(v[3])};
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
          }
        inv = rsqrtf(wave_sum(ss) / (float)a.K + a.eps);
      }
#pragma unroll
      for (int s_ = 0; s_ < XA; ++s_)
        if (s_ / cpl == rw) {
          const int cl = s_ % cpl, c = lane + 64 * cl;
          u32x4_t v = xr[s_];
          if (PRO == MPRO_RMSNORM) {
            u32x4_t wu = nw[0];
#pragma unroll
            for (int q = 1; q < NWC; ++q) wu = (cl == q) ? nw[q] : wu;
            u32x4_t o;
            o[0] = pack_bf2(bf_lo(wu[0]) * rbf(bf_lo(v[0]) * inv), bf_hi(wu[0]) * rbf(bf_hi(v[0]) * inv));
            o[1] = pack_bf2(bf_lo(wu[1]) * rbf(bf_lo(v[1]) * inv), bf_hi(wu[1]) * rbf(bf_hi(v[1]) * inv));
            o[2] = pack_bf2(bf_lo(wu[2]) * rbf(bf_lo(v[2]) * inv), bf_hi(wu[2]) * rbf(bf_hi(v[2]) * inv));
            o[3] = pack_bf2(bf_lo(wu[3]) * rbf(bf_lo(v[3]) * inv), bf_hi(wu[3]) * rbf(bf_hi(v[3]) * inv));
            v = o;
          }
          if (m < a.M && c < nch) *reinterpret_cast<u32x4_t*>(smem + (size_t)m * P + (size_t)c * 16) = x_order(v);
        }
    }
    __syncthreads();
  } else if constexpr (XS < 0) {
    // row-trip form (any row count, K <= 64 * 8 * CP): wave w owns batch rows w, w + 4, ..; per trip it loads RPT of its rows
    // whole (CP chunks per lane and row, all loads of the trip in flight together), takes their RMS statistics from the
    // registers and writes them normalised to LDS - the loop form above costs one dependent L2 round trip per chunk and
    // row (56 of them at K = 3584 and 16 rows).  The first unit's weights go out behind the LAST trip's loads.
    constexpr int CP = XS == -2 ? 3 : 8, RPT = XS == -2 ? 4 : 2;
    const int nch = kx >> 3, k0 = kb_fix * 128;
    const int ntrip = (((a.M + 3) >> 2) + RPT - 1) / RPT;
    u32x4_t nw[PRO == MPRO_RMSNORM ? CP : 1];
    if (PRO == MPRO_RMSNORM) {
#pragma unroll
      for (int j = 0; j < CP; ++j) nw[j] = *reinterpret_cast<const u32x4_t*>(a.norm_w + min(k0 + min(lane + 64 * j, nch - 1) * 8, a.K - 8));
    }
    auto trip = [&](int t, auto last) __attribute__((always_inline)) {
      u32x4_t xr[RPT][CP];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const bf16_t* xrow = a.x + (size_t)min(wave + 4 * (t * RPT + r), a.M - 1) * a.ldx;
#pragma unroll
        for (int j = 0; j < CP; ++j) xr[r][j] = *reinterpret_cast<const u32x4_t*>(xrow + min(k0 + min(lane + 64 * j, nch - 1) * 8, a.K - 8));
      }
      __builtin_amdgcn_sched_barrier(0);
      // behind the small loads (they return first); the 7-chunk register set goes out after the prologue instead: together
      // with a trip's rows it would cost the second workgroup of a CU
      if (decltype(last)::value && NCW == 3) load_w(min(u, n_units - 1), wvA);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const int m = wave + 4 * (t * RPT + r);
        float inv = 1.f;
        if (PRO == MPRO_RMSNORM) {
          float ss = 0.f;
#pragma unroll
          for (int j = 0; j < CP; ++j) {
            const bool in = lane + 64 * j < nch;           // chunks past the row count as zeros: ss + 0 * 0 is exact
            const u32x4_t v = xr[r][j];
            const unsigned w0 = in ? v[0] : 0u, w1 = in ? v[1] : 0u, w2 = in ? v[2] : 0u, w3 = in ? v[3] : 0u;
            const float f[8] = {bf_lo(w0), bf_hi(w0), bf_lo(w1), bf_hi(w1), bf_lo(w2), bf_hi(w2), bf_lo(w3), bf_hi(w3)};
#pragma unroll
            for (int q = 0; q < 8; ++q) ss += f[q] * f[q];
          }
          inv = rsqrtf(wave_sum(ss) / (float)a.K + a.eps);
        }
#pragma unroll
        for (int j = 0; j < CP; ++j) {
          const int c = lane + 64 * j;
          u32x4_t v = xr[r][j];
          if (PRO == MPRO_RMSNORM) {
            const u32x4_t wu = nw[PRO == MPRO_RMSNORM ? j : 0];
            u32x4_t o;
            o[0] = pack_bf2(bf_lo(wu[0]) * rbf(bf_lo(v[0]) * inv), bf_hi(wu[0]) * rbf(bf_hi(v[0]) * inv));
            o[1] = pack_bf2(bf_lo(wu[1]) * rbf(bf_lo(v[1]) * inv), bf_hi(wu[1]) * rbf(bf_hi(v[1]) * inv));
            o[2] = pack_bf2(bf_lo(wu[2]) * rbf(bf_lo(v[2]) * inv), bf_hi(wu[2]) * rbf(bf_hi(v[2]) * inv));
            o[3] = pack_bf2(bf_lo(wu[3]) * rbf(bf_lo(v[3]) * inv), bf_hi(wu[3]) * rbf(bf_hi(v[3]) * inv));
            v = o;
          }
          if (m < a.M && c < nch) *reinterpret_cast<u32x4_t*>(smem + (size_t)m * P + (size_t)c * 16) = x_order(v);
        }
      }
    };
    for (int t = 0; t + 1 < ntrip; ++t) trip(t, std::false_type{});
    trip(ntrip - 1, std::true_type{});
    if (NCW != 3) load_w(min(u, n_units - 1), wvA);
    __syncthreads();
  } else {
    stage_x(kb_fix);
    load_w(min(u, n_units - 1), wvA);
    __syncthreads();
  }
  group_sums();

  // epilogue of one finished 16 x 16 tile: thread tid holds element (n_l = tid >> 4, m = tid & 15); called by the whole workgroup
  auto finish = [&](int tile, float v) __attribute__((always_inline)) {
    const int n_l = tid >> 4, m = tid & 15;
    const int n = tile_row<EPI>(a, tile, n_l);
    if (EPI == MEPI_ROPE_KV || (EPI & VLM_EPI_SWIGLU)) {
      __syncthreads();                  // (uniform: `mine` is per workgroup) every thread has read its part[] sums
      part[tid] = v;
      __syncthreads();
    }
    if (EPI == MEPI_ROPE_KV) {
      const int half = a.rk.D >> 1, tph = half >> 3, n_rot = (a.rk.Hq + a.rk.Hkv) * tph;
      // SuScaledRoPE's per-call rule (rope_utils.py:168-172): long factors for every row of the step once ANY row's
      // cache offset has reached original_max (scalar loop over the <= 16 rows: uniform, only when the model has two tables)
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
        a.y[a.y_tiled ? ((size_t)(n >> 4) * 16 + m) * 8 + ((n >> 1) & 7) : (size_t)m * a.ldy + (n >> 1)] = f2bf(swiglu_(rbf(v), rbf(part[tid + 16])));
    } else if (m < a.M && n < a.N) {
      if (EPI & VLM_EPI_BIAS) v = (W4 ? rbf(v) : v) + bf2f(a.bias[n]);
      if (EPI & VLM_EPI_RESIDUAL) v = rbf(v) + bf2f(a.res[(size_t)m * a.ldres + n]);
      a.y[a.y_tiled ? ((size_t)(n >> 3) * 16 + m) * 8 + (n & 7) : (size_t)m * a.ldy + n] = f2bf(v);
    }
  };

  // one unit: transposition + MFMAs of this wave's chunks, cross-wave / cross-workgroup reduction, epilogue
  auto unit = [&](int u, u32x4_t (&wv)[NCW][4]) __attribute__((always_inline)) {
    const int tile = u / a.KS, ks = u % a.KS;
    const int kb0 = ks * a.bpk, kb1 = min(a.nblk, kb0 + a.bpk);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const char* xrow0 = smem + (size_t)mrow * P - (FULLX ? 0 : (size_t)kb0 * 256);
    const char* xrow = xrow0 + g * 16;
    // one chunk of this wave: W4 - registers only; bf16 - through the wave's transposition region
    auto chunk = [&](int i, int b) __attribute__((always_inline)) {
      if (W4) {
        const u32x4_t w = wv[i][0];
        const bool lo = lane < 32;
        // lanes l and l ^ 32 (same row, quarters g and g ^ 2) trade words so that words 0 / 1 of every lane come from the
        // chunk's first 64-wide group and words 2 / 3 from the second: the lower half sends its words 2 / 3, the upper 0 / 1
        const unsigned s0 = __shfl_xor(lo ? w[2] : w[0], 32), s1 = __shfl_xor(lo ? w[3] : w[1], 32);
        const unsigned v4[4] = {lo ? w[0] : s0, lo ? w[1] : s1, lo ? s0 : w[2], lo ? s1 : w[3]};
        unsigned* sb_i = sbr + i * 32;
        if (g == 0) *reinterpret_cast<uint2*>(sb_i + r16 * 2) = uint2{wv[i][1][0], wv[i][1][1]};
        const float* xs_m = xsum + (size_t)mrow * (kx >> 6) + (size_t)(b - (FULLX ? 0 : kb0)) * 2;
        // k offsets (bytes into the chunk's 256 B of x) of the lane's words: own quarter 32 g, the partner's 32 (g ^ 2)
        const char* xb = xrow0 + (size_t)b * 256;
        const int off[2] = {lo ? 64 * g : 64 * (g - 2) + 32, lo ? 64 * (g + 2) : 64 * g + 32};
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          f32x4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const unsigned wd = v4[2 * gq + k2];
            const u32x4_t af = {(wd & 0x000F000Fu) | 0x43004300u, ((wd >> 4) & 0x000F000Fu) | 0x43004300u,
                                ((wd >> 8) & 0x000F000Fu) | 0x43004300u, ((wd >> 12) & 0x000F000Fu) | 0x43004300u};
            const u32x4_t bf = *reinterpret_cast<const u32x4_t*>(xb + off[gq] + 16 * k2);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bf), d, 0, 0, 0);
          }
          const float sx = xs_m[gq];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned sbw = sb_i[(4 * g + q) * 2 + gq];
            acc[q] += bf_lo(sbw) * (d[q] - 128.f * sx) + bf_hi(sbw) * sx;
          }
        }
      } else {
        // chunk -> the wave's region [16 rows][272 B] (as loaded: row 4 j + g, byte 16 r16), then the fragments
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4_t*>(wreg + (4 * j + g) * 272 + r16 * 16) = wv[i][j];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const u32x4_t af = *reinterpret_cast<const u32x4_t*>(wreg + r16 * 272 + kb * 64 + g * 16);
          const u32x4_t bf = *reinterpret_cast<const u32x4_t*>(xrow + (size_t)b * 256 + kb * 64);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bf), acc, 0, 0,
                                                        0);
        }
      }
    };
    // W4: the wave's valid chunks (wave-uniform count) as straight-line code - no per-chunk branch, so the MFMAs,
    // dequantisation and LDS reads of neighbouring chunks overlap; the common counts NCW, NCW - 1, NCW - 2 get a copy each
    const int n_valid = max(0, min(NCW, (kb1 - kb0 - wave + 3) >> 2));
    auto straight = [&](auto nc) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < decltype(nc)::value; ++i) chunk(i, kb0 + wave + 4 * i);
    };
    if (n_valid == NCW) {
      straight(std::integral_constant<int, NCW>{});
    } else if (W4 && NCW > 1 && n_valid == NCW - 1) {
      straight(std::integral_constant<int, (NCW > 1 ? NCW - 1 : 1)>{});
    } else if (W4 && NCW > 2 && n_valid == NCW - 2) {
      straight(std::integral_constant<int, (NCW > 2 ? NCW - 2 : 1)>{});
    } else {
#pragma unroll
      for (int i = 0; i < NCW; ++i) {
        const int b = kb0 + wave + 4 * i;
        if (b < kb1) chunk(i, b);                             // wave-uniform
      }
    }
    // this set is free again: the unit NSETS ahead goes out now.  (A slot the unit did not multiply - a wave's surplus chunks
    // of a short segment - still has its load pending as far as the compiler knows; the empty use retires it here.)
#pragma unroll
    for (int i = 0; i < NCW; ++i)
#pragma unroll
      for (int j = 0; j < (W4 ? 2 : 4); ++j) asm volatile("" ::"v"(wv[i][j]));
    if (u + NSETS * G < n_units) load_w(u + NSETS * G, wv);

    // D[n = 4 g + i][m = r16]  ->  part[wave][n * 16 + m]
#pragma unroll
    for (int i = 0; i < 4; ++i) part[wave * 256 + (4 * g + i) * 16 + r16] = acc[i];
    __syncthreads();
    const int n_l = tid >> 4, m = tid & 15;
    float v = (part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid]);
    if (a.KS > 1) {
      // partial tile -> workspace with agent-scope (L2-write-through) stores; the reader uses agent-scope loads.  Tickets and
      // merges come after the workgroup's LAST unit (below): no unit waits for a hand-off (the first version did all of it
      // after every unit: store -> vmcnt(0), which also drains the next unit's weight loads -> ticket -> partial loads).
      // (An agent-scope release FENCE instead of the write-through stores writes back the whole XCD L2 - 20 us per launch.)
      __hip_atomic_store(a.ws + (size_t)u * 256 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      finish(tile, v);
    }
    __syncthreads();                      // part[] is rewritten by the next unit
  };

#pragma unroll
  for (int s_ = 1; s_ < NSETS; ++s_)
    if (u + s_ * G < n_units) load_w(u + s_ * G, wv[s_]);
  while (u < n_units) {
#pragma unroll
    for (int s_ = 0; s_ < NSETS; ++s_) {
      if (u < n_units) unit(u, wv[s_]);
      u += G;
    }
  }
  if (a.KS > 1) {
    // deferred hand-off: all partial tiles of this workgroup are on their way (agent-scope stores); ONE wait, then the
    // tickets of all its units at once (thread i: unit blockIdx.x + i G), then the merges of the tiles it arrived last at -
    // partials summed in the fixed order ks = 0 .. KS - 1, so the result does not depend on who merges
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flags = reinterpret_cast<int*>(red);
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
        v += __hip_atomic_load(a.ws + ((size_t)tile_i * a.KS + k) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // every segment of this tile has arrived: re-arm (launches that share the array are ordered by their stream)
      if (tid == 0) __hip_atomic_store(a.tickets + tile_i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      finish(tile_i, v);
      __syncthreads();
    }
  }
}

template <int PRO, int EPI, bool FULLX, int NCW, int XS, bool W4>
int mfma_launch2(const MfmaArgs& a, size_t lds, int n_units, hipStream_t st) {
  auto kern = gemv_mfma_kernel<PRO, EPI, FULLX, NCW, XS, W4>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return VLM_ERR_HIP + (int)e;
  }
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
  static const int wgs_per_cu = [] { const char* e = getenv("VLM_GEMV_MFMA_WGS_PER_CU"); return e ? max(1, min(4, atoi(e))) : 2; }();
  int grid = min(n_units, 256 * min(nb, wgs_per_cu));   // few resident workgroups: the activation staging is paid per workgroup
  grid -= grid % a.KS;                             // a workgroup keeps its K segment (unit id = tile * KS + ks)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

template <int PRO, int EPI, bool FULLX, bool W4>
int mfma_launch1(const MfmaArgs& a, size_t lds, int n_units, hipStream_t st) {
  const bool ncw3 = a.bpk <= 12;
  const int kx = FULLX ? a.K : a.bpk * 128;
  const int cpl = ((kx >> 3) + 63) / 64, slots = ((a.M + 3) / 4) * cpl;
  constexpr int S6 = 6, S14 = 14;
  // activation prologue: S6 / S14 = everything in registers at once; T3 / T8 = the row-trip form (K <= 1536 / <= 4096 staged
  // elements per row); 0 = the loop form (longer rows: none of the models here)
  constexpr int T3 = -2, T8 = -1;
  static const bool trips = [] { const char* e = getenv("VLM_GEMV_MFMA_TRIPS"); return !e || atoi(e) != 0; }();   // A/B knob
  if (ncw3) {
    if (cpl <= 3 && slots <= 6) return mfma_launch2<PRO, EPI, FULLX, 3, S6, W4>(a, lds, n_units, st);
    if constexpr (PRO == MPRO_NONE)
      if (slots <= 14) return mfma_launch2<PRO, EPI, FULLX, 3, S14, W4>(a, lds, n_units, st);
    if (trips && cpl <= 3) return mfma_launch2<PRO, EPI, FULLX, 3, T3, W4>(a, lds, n_units, st);
    if (trips && cpl <= 8) return mfma_launch2<PRO, EPI, FULLX, 3, T8, W4>(a, lds, n_units, st);
    return mfma_launch2<PRO, EPI, FULLX, 3, 0, W4>(a, lds, n_units, st);
  }
  if (cpl <= 3 && slots <= 6) return mfma_launch2<PRO, EPI, FULLX, 7, S6, W4>(a, lds, n_units, st);
  if (trips && cpl <= 3) return mfma_launch2<PRO, EPI, FULLX, 7, T3, W4>(a, lds, n_units, st);
  if (trips && cpl <= 8) return mfma_launch2<PRO, EPI, FULLX, 7, T8, W4>(a, lds, n_units, st);
  return mfma_launch2<PRO, EPI, FULLX, 7, 0, W4>(a, lds, n_units, st);
}

template <int PRO, int EPI, bool FULLX>
int mfma_launch(const MfmaArgs& a, size_t lds, int n_units, hipStream_t st) {
  // the bf16 and the 4-bit instantiations are two translation units (gemv_mfma_w4.hip includes this file with
  // VLM_MFMA_W4_TU defined): ~120 kernels each, compiled in parallel
#ifdef VLM_MFMA_W4_TU
  return a.Wsb ? mfma_launch1<PRO, EPI, FULLX, true>(a, lds, n_units, st) : -1;
#else
  return a.Wsb ? -1 : mfma_launch1<PRO, EPI, FULLX, false>(a, lds, n_units, st);
#endif
}

}  // namespace

// -> VLM_OK, an error, or -1: shape not handled here (the caller takes the v_dot2c kernels)
static int mfma_try(const void* x, const void* W, const void* Wsb, const void* bias, const void* res, const void* norm_w, void* y,
                    int M, int N, int K, int ldx, int ldw, int ldy, int ldres, float eps, int epilogue, const VlmRopeKv* rk,
                    void* ws, void* stream);

#ifndef VLM_MFMA_W4_TU
// 4096 units of partials + tickets + the normalised activation rows of the second form (gemv_mfma2.hip)
VLM_INTERNAL size_t vlm_gemv_mfma_ws_bytes(void) { return VLM_MFMA_WS_XN_OFFSET + VLM_MFMA_WS_XN_BYTES; }

VLM_INTERNAL int vlm_gemv_mfma_try(const void* x, const void* W, const void* bias, const void* res, const void* norm_w, void* y, int M, int N,
                                   int K, int ldx, int ldw, int ldy, int ldres, float eps, int epilogue, const VlmRopeKv* rk,
                                   void* ws, void* stream) {
  return mfma_try(x, W, nullptr, bias, res, norm_w, y, M, N, K, ldx, ldw, ldy, ldres, eps, epilogue, rk, ws, stream);
}

#else
// the same over MLX affine 4-bit weights (Wq words [N][K/8], Wsb (scale | bias << 16) [N][K/64]): all supported row counts
// (the v_dot2c 4-bit GEMVs stop at 8 rows)
VLM_INTERNAL int vlm_gemv_mfma_try_w4(const void* x, const void* Wq, const void* Wsb, const void* bias, const void* res,
                                      const void* norm_w, void* y, int M, int N, int K, int ldx, int ldy, int ldres, float eps,
                                      int epilogue, const VlmRopeKv* rk, void* ws, void* stream) {
  if (!Wsb) return -1;
  return mfma_try(x, Wq, Wsb, bias, res, norm_w, y, M, N, K, ldx, 8, ldy, ldres, eps, epilogue, rk, ws, stream);
}

#endif

static int mfma_try(const void* x, const void* W, const void* Wsb, const void* bias, const void* res, const void* norm_w, void* y,
                    int M, int N, int K, int ldx, int ldw, int ldy, int ldres, float eps, int epilogue, const VlmRopeKv* rk,
                    void* ws, void* stream) {
  // A/B knobs: VLM_GEMV_MFMA=0 turns the path off, VLM_GEMV_MFMA_MIN_M moves the row count it starts at.  Default 5: a
  // step costs the same here for 4, 8 or 16 rows (1.45 ms at 2B dims) while the v_dot2c step grows with the rows (1.25 ms at
  // 4, 1.8 ms at 8): profiles/r02_mfma_gemv.txt.  The qkv + RoPE + KV-write form starts at 9 rows (6.2 vs 7.2 us at 8).
  static const bool enabled = [] { const char* e = getenv("VLM_GEMV_MFMA"); return !e || atoi(e) != 0; }();
  static const int min_m = [] { const char* e = getenv("VLM_GEMV_MFMA_MIN_M"); return e ? atoi(e) : 5; }();
  const bool y_tiled = (epilogue & VLM_EPI_Y_TILED) != 0;      // (this form's epilogue writes it; the other forms are skipped)
  epilogue &= ~VLM_EPI_Y_TILED;
  if (!enabled || M < (rk ? max(min_m, 9) : min_m) || M > 16 || K % 128 || ldx % 8 || ldw % 8) return -1;
  const bool rope = rk != nullptr;
  if (y_tiled && (rope || Wsb)) return -1;
  if (rope && (rk->D % 16 || !norm_w || !bias)) return -1;
  if (!rope && epilogue != VLM_EPI_NONE && epilogue != VLM_EPI_BIAS && epilogue != VLM_EPI_RESIDUAL && epilogue != VLM_EPI_SWIGLU &&
      epilogue != (VLM_EPI_BIAS | VLM_EPI_RESIDUAL))
    return -1;
  if (norm_w && (epilogue & VLM_EPI_RESIDUAL)) return -1;
  if ((epilogue & VLM_EPI_SWIGLU) && (N % 16)) return -1;
#ifndef VLM_MFMA_W4_TU
  if (!Wsb && !norm_w && !rope && !y_tiled) {
    // few row tiles x long K (the down projections): one workgroup per tile, K split over its 16 waves, no cross-workgroup
    // hand-off (gemv_mfma_longk.hip)
    const int rc3 = vlm_gemv_mfma_longk_try(x, W, bias, res, y, M, N, K, ldx, ldw, ldy, ldres, epilogue, stream);
    if (rc3 != -1) return rc3;
  }
#endif
  {
    // the second form first (gemv_mfma2.hip: activations in registers, two workgroups per CU); VLM_GEMV_MFMA2=0: A/B knob
    static const bool v2 = [] { const char* e = getenv("VLM_GEMV_MFMA2"); return !e || atoi(e) != 0; }();
    if (v2 && !y_tiled) {
#ifdef VLM_MFMA_W4_TU
      const int rc2 = Wsb ? vlm_gemv_mfma2_try_w4(x, W, Wsb, bias, res, norm_w, y, M, N, K, ldx, ldw, ldy, ldres, eps, epilogue, rk, ws, stream) : -1;
#else
      const int rc2 = Wsb ? -1 : vlm_gemv_mfma2_try_bf16(x, W, Wsb, bias, res, norm_w, y, M, N, K, ldx, ldw, ldy, ldres, eps, epilogue, rk, ws, stream);
#endif
      if (rc2 != -1) return rc2;
    }
  }
  MfmaArgs a{};
  a.x = (const bf16_t*)x; a.W = (const bf16_t*)W; a.bias = (const bf16_t*)bias; a.res = (const bf16_t*)res;
  a.norm_w = (const bf16_t*)norm_w; a.y = (bf16_t*)y; a.Wsb = (const unsigned*)Wsb;
  a.M = M; a.N = N; a.K = K; a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldres = ldres; a.eps = eps;
  a.y_tiled = y_tiled ? 1 : 0;
  if (rope) {
    a.rk = *rk;
    if (N != (rk->Hq + 2 * rk->Hkv) * rk->D) return -1;
    a.n_tiles = (rk->Hq + rk->Hkv) * (rk->D / 16) + rk->Hkv * rk->D / 16;
  } else {
    a.n_tiles = vlm_cdiv(N, 16);
  }
  a.nblk = K / 128;
  // K segments.  A workgroup streams its 16-row x segment slice at a few tens of bytes per clock, so a unit of <= ~48 KB
  // (12 chunks: the double-buffered form) is the target and splitting further only adds the cross-workgroup hand-off
  // (qkv at KS = 3: 10 us, one dependent chain of partial store -> ticket -> partial loads -> epilogue)
  int KS = vlm_cdiv(a.nblk, 4 * NCW);
  // ... and only while the row tiles alone do not fill the chip: every unit of a split tile ends in the hand-off chain
  // (partial store -> vmcnt(0), which also drains the next unit's weight loads -> ticket -> partial loads), ~5-10 us per
  // unit with one workgroup per CU.  Measured at Phi-3.5 dims (K = 3072, 1024 / 768 row tiles, 16 rows): gate/up and qkv
  // 48 us at KS = 2 (profiles/r02_phi35v_kernel_stats.txt)
  // Measured per projection at 16 rows (profiles/r03_mfma_shapes.txt, scripts/mfma_shapes.py; the knobs below are its A/B handles):
  //  - norm-prologue forms (every workgroup stages ALL of x: one workgroup per CU at K >= 3072) never gain from a K split:
  //    7B qkv 23.9 -> 18.1 us, Phi-3.5 4-bit qkv 24.4 -> 19.9 us without it;
  //  - forms without a norm: long segments (up to 28 chunks = the 7-chunk kernel, x segment up to 115 KB: one workgroup per
  //    CU) win when they remove the split altogether (7B o_proj 12.7 -> 9.7 us, Phi-3.5 o_proj 11.7 -> 9.4) or when there
  //    are still >= 768 units (7B down 49.9 -> 43.3 us, Mistral down 40.0 -> 32.4); with few row tiles the short segments
  //    keep more workgroups busy (2B down: 14.7 us at 6 x 12 chunks, 16.1 at 3 x 24).
  static const int seg_env = [] { const char* e = getenv("VLM_GEMV_MFMA_SEG_CHUNKS"); return e ? max(4, min(28, atoi(e))) : 0; }();
  static const int norm_split = [] { const char* e = getenv("VLM_GEMV_MFMA_NORM_SPLIT"); return e ? atoi(e) : 0; }();
  static const int lds_cap_kb = [] { const char* e = getenv("VLM_GEMV_MFMA_LDS_KB"); return e ? max(48, min(160, atoi(e))) : 160; }();
  int seg_chunks = seg_env ? seg_env : 12;
  if (!seg_env && !norm_w) {
    const int ks28 = vlm_cdiv(a.nblk, 4 * NCW);
    if (ks28 == 1 || a.n_tiles * ks28 >= 768) seg_chunks = 4 * NCW;
  }
  if (ws)
    while (vlm_cdiv(a.nblk, KS) > (norm_w ? 12 : seg_chunks) && KS < 16 && (size_t)a.n_tiles * (KS + 1) <= 4096 &&
           (!norm_w || a.n_tiles * KS < norm_split))      // (the segment forms without a norm also split to fit their x segment in LDS)
      ++KS;
  if (KS > 1 && (!ws || a.n_tiles > 8192 || (size_t)a.n_tiles * KS > 4096)) return -1;
  a.KS = KS;
  a.bpk = vlm_cdiv(a.nblk, KS);
  if (a.bpk > 4 * NCW) return -1;
  a.ws = (float*)ws;
  a.tickets = ws ? (unsigned*)((char*)ws + (size_t)4096 * 256 * 4) : nullptr;
  const int n_units = a.n_tiles * KS;
  const size_t tail = 4 * WREG + 4096 + 256 + 64 + 4 * 7 * 128 + (Wsb ? (size_t)M * (K / 64) * 4 : 0);      // + sbr, xsum (4-bit)
  const size_t lds_full = (size_t)M * ((size_t)K * 2 + 16) + tail, lds_seg = (size_t)M * ((size_t)a.bpk * 256 + 16) + tail;
  const bool fullx = lds_full <= (size_t)(norm_w ? 160 : 100) * 1024;      // (one workgroup per CU above 80 KB: only when the norm needs it)
  if (!fullx && (norm_w || lds_seg > (size_t)lds_cap_kb * 1024)) return -1;
  static const bool debug = [] { const char* e = getenv("VLM_GEMV_MFMA_DEBUG"); return e && atoi(e) != 0; }();
  if (debug)
    fprintf(stderr, "[gemv_mfma] M=%d N=%d K=%d %s%s epi=%d: tiles=%d KS=%d bpk=%d units=%d %s lds=%zu KB\n", M, N, K, Wsb ? "w4 " : "",
            norm_w ? "norm" : "plain", rope ? -1 : epilogue, a.n_tiles, KS, a.bpk, n_units, fullx ? "fullx" : "seg",
            (fullx ? lds_full : lds_seg) >> 10);
  hipStream_t st = (hipStream_t)stream;
#define GO(P, E) return fullx ? mfma_launch<P, E, true>(a, lds_full, n_units, st) : mfma_launch<P, E, false>(a, lds_seg, n_units, st)
#define GOF(P, E) return mfma_launch<P, E, true>(a, lds_full, n_units, st)
  if (rope) GOF(MPRO_RMSNORM, MEPI_ROPE_KV);
  if (norm_w) {
    switch (epilogue) {
      case VLM_EPI_NONE: GOF(MPRO_RMSNORM, VLM_EPI_NONE);
      case VLM_EPI_BIAS: GOF(MPRO_RMSNORM, VLM_EPI_BIAS);
      case VLM_EPI_SWIGLU: GOF(MPRO_RMSNORM, VLM_EPI_SWIGLU);
      default: return -1;
    }
  }
  switch (epilogue) {
    case VLM_EPI_NONE: GO(MPRO_NONE, VLM_EPI_NONE);
    case VLM_EPI_BIAS: GO(MPRO_NONE, VLM_EPI_BIAS);
    case VLM_EPI_RESIDUAL: GO(MPRO_NONE, VLM_EPI_RESIDUAL);
    case VLM_EPI_BIAS | VLM_EPI_RESIDUAL: GO(MPRO_NONE, VLM_EPI_BIAS | VLM_EPI_RESIDUAL);
    case VLM_EPI_SWIGLU: GO(MPRO_NONE, VLM_EPI_SWIGLU);
    default: return -1;
  }
#undef GO
#undef GOF
}
