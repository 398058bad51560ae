// Flash-attention forward for prefill on gfx950: variable-length segments
// (cu_seqlens), optional causal mask, GQA, head_dim 80 (ViT) or 128 (LLM).
//
// Replaces mx.fast.scaled_dot_product_attention on the prefill path:
//   * ViT, mask=None, one call per cu_seqlens segment in a Python loop
//     (reference mlx_vlm/models/qwen2_vl/vision.py:148-158)  -> one launch here
//   * LLM prompt, mask="causal" (reference mlx_vlm/models/base.py:214-228,
//     366-373; mlx_vlm/models/qwen2_vl/language.py:115-118)
// fp32 scores / softmax statistics / accumulation as MLX does; P is rounded to
// bf16 for the P.V MFMA (flash-attention convention; tolerance stated in tests).
//
// CDNA4 mapping: 4 waves x 32 query rows per workgroup, 64-key K/V tiles in LDS.
// Both products are issued TRANSPOSED on v_mfma_f32_32x32x16_bf16,
//     S^T = K . Q^T      (D[i=key][j=q])      O^T = V^T . P^T   (D[i=d][j=q])
// so the C/D column index is the query row for both: every per-query quantity
// (running max, sum, rescale factor) is lane-local (q = lane & 31), the softmax
// needs only one cross-lane op per statistic (lane ^ 32 holds the other half of
// the keys), and the exponentiated S^T registers ARE the B operand of the second
// MFMA with no cross-lane movement: the 8 k-slots a lane feeds to MFMA #m of a
// 32-key block are its registers 8m..8m+7, i.e. keys 16m+4h+{0..3} and
// 16m+8+4h+{0..3} (h = lane>>5).  V stays ROW-MAJOR in LDS (straight 16-byte copies of the
// [key][d] rows, like K) and the V^T fragments come out of gfx950's transposing LDS read
// (ds_read_b64_tr_b16: in a 16-lane group lane i passes the address of row i/4, columns
// 4 (i%4)..+3 of a [4 keys][16 d] block and receives column i of its 4 rows - measured with
// scripts/tr_probe.hip), two per fragment with exactly that key permutation.  K rows are padded by
// 16 B; the V row pitch is 96 / 160 elements (pitch in dwords = 16 or 48 mod 64: the 4 rows a
// 32-lane half reads fall on disjoint banks).  The next K/V tile's global loads are issued before the
// MFMAs of the current one (register staging, T14-style) and written to LDS after
// the barrier.  Output is stored as 8-byte bf16x4 pieces (4 consecutive d).
#include <stdlib.h>

#include "common.hpp"
#include "../../include/vlm_hip.h"

#ifdef ATTN_STAMPS
// measurement build only (scripts/attn_probe.hip): per-phase cycle sums of one workgroup, kept in SGPRs inside the loop
__device__ unsigned long long g_attn_stamps[4][8];
#define ST_DECL unsigned long long st_prev = 0, st_sum[6] = {0, 0, 0, 0, 0, 0}; const bool st_on = blockIdx.x == ATTN_STAMPS && blockIdx.y == 0
#define ST_BEGIN() do { if (st_on) { __builtin_amdgcn_sched_barrier(0); st_prev = __builtin_amdgcn_s_memtime(); } } while (0)
#define ST_MARK(i) do { if (st_on) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[i] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define ST_DECL
#define ST_BEGIN()
#define ST_MARK(i)
#endif

namespace {

constexpr int BQ = 128;   // query rows per workgroup (4 waves x 32)
constexpr int BKV = 64;   // keys per tile
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

template <int D, bool CAUSAL, bool HILO = false>
// HILO: P as two bf16 MFMA operands (hi = bf16(p), lo = bf16(p - hi)): 16 mantissa bits of p reach O^T, as in the decode kernels
// (attn_pagesplit.hpp) - here at the price of a second P.V MFMA per fragment in a kernel whose matrix pipe is half busy.
// (D <= 80: three waves per SIMD = three workgroups per CU - 168 registers, the only spills are epilogue values saved before
//  the loop; the D = 64 form had three waves all along and ran at 516 TF where this one ran at 440)
__global__ __launch_bounds__(256, (D <= 80 ? 3 : 2)) void attn_prefill_kernel(
    const bf16_t* __restrict__ qp, const bf16_t* __restrict__ kp, const bf16_t* __restrict__ vp, bf16_t* __restrict__ op,
    int q_stride, int k_stride, int v_stride, int o_stride, const int* __restrict__ cu, int nseg, int Hq, int Hkv,
    float scale_log2, int uniform_nqb, const int* __restrict__ qstart) {
  constexpr int DP = (D + 31) / 32 * 32;   // padded head dim for the O^T tiles
  constexpr int NKS = D / 16;              // k-steps of the S^T product
  constexpr int NDB = DP / 32;             // 32-wide d blocks of O^T
  constexpr int K_LD = D + 8;              // padded K row (elements)
  constexpr int KCH = BKV * D / 8;         // 16-byte chunks in a K (or V) tile
  constexpr int K_PER = (KCH + 255) / 256;
  constexpr int V_LD = (DP % 128 == 32 || DP % 128 == 96) ? DP : DP + 32;   // V row pitch: 96 (D 64, 80), 160 (D 128)
  // a spare column of the padded head dim (D = 80 in 96) holds 1.0 in the V tile: the row sums l come out of the P.V MFMAs as
  // row D of O^T (32 adds per tile and lane less; numerator and denominator sum the SAME bf16-rounded P)
  constexpr bool ONES = DP > D;
  constexpr int LDB = D / 32, LR = ((D % 32) / 8) * 4;     // ... register ot[LDB][LR] of the lanes with h == 0

  // double-buffered: tile t+1 is written while the other waves may still read tile t -> ONE barrier per tile
  __shared__ __attribute__((aligned(16))) bf16_t Ks2[2][BKV * K_LD];
  __shared__ __attribute__((aligned(16))) bf16_t Vs2[2][BKV * V_LD];

  // ---- locate (segment, head, q block) ----
  int seg = 0, qb = 0, head = blockIdx.y;
  if (uniform_nqb > 0) {
    // every segment has uniform_nqb query blocks and nseg * Hq % 8 == 0 (checked on the host): workgroups are
    // dispatched round-robin over the 8 XCDs, so give all query blocks of one (segment, head) the same
    // id % 8 - its K/V rows are then fetched into ONE XCD's L2 instead of up to uniform_nqb of them
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, i = lin >> 3;
    const int pair = xcd + 8 * (i / uniform_nqb);
    qb = i % uniform_nqb;
    seg = pair / Hq;
    head = pair % Hq;
  } else {
    // qstart (round 6): rows of segment s before qstart[s] are KEYS ONLY - the cached prefix of a prompt chunk appended to a
    // non-empty cache (LanguageModel._prefill_onto_cache).  The segment's query blocks start at that row; masks and the causal
    // end work on absolute rows, so nothing else changes (the prefix rows used to carry zero queries: O(Tf^2) for a chunk).
    int bid = blockIdx.x;
    for (; seg < nseg; ++seg) {
      const int nb = (cu[seg + 1] - cu[seg] - (qstart ? qstart[seg] : 0) + BQ - 1) / BQ;
      if (bid < nb) { qb = bid; break; }
      bid -= nb;
    }
    if (seg >= nseg) return;
  }
  const int seg_start = cu[seg], seg_len = cu[seg + 1] - seg_start;
  const int qbase = (qstart ? qstart[seg] : 0) + qb * BQ;        // first row of this query block inside the segment
  const int kvh = head / (Hq / Hkv);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  __builtin_assume(tid >= 0 && tid < 256);
  const int qrow = qbase + wave * 32 + (lane & 31);            // row inside the segment
  const int qrow_c = min(qrow, seg_len - 1);
  const bool wave_rows = qbase + wave * 32 < seg_len;            // wave-uniform

  // ---- Q fragments (B operand of S^T): q = lane&31, d = ks*16 + 8h .. +8 ----
  bf16x8_t qf[NKS];
  {
    const bf16_t* qr = qp + (size_t)(seg_start + qrow_c) * q_stride + (size_t)head * D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qr + ks * 16);
  }

  const int kv_end = CAUSAL ? min(seg_len, qbase + BQ) : seg_len;
  const int ntiles = (kv_end + BKV - 1) / BKV;
  const bf16_t* kbase = kp + (size_t)seg_start * k_stride + (size_t)kvh * D;
  const bf16_t* vbase = vp + (size_t)seg_start * v_stride + (size_t)kvh * D;

  // staging registers: the K and V tiles together are 2 KCH 16-byte chunks = NCH per thread (D = 80: 5, not 3 + 3 - four
  // registers that decide between two and three waves per SIMD); chunk g = tid + 256 i is a K chunk below KCH, a V chunk from
  // there on (wave-uniform: KCH is a multiple of 128)
  constexpr int NCH = 2 * KCH / 256;
  static_assert(2 * KCH % 256 == 0 && KCH % 128 == 0, "tile chunks must divide over the 256 threads");
  u32x4_t rs[NCH];
  auto gload = [&](int t) {
    const int j0 = t * BKV;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int g = tid + 256 * i;
      const bool isk = (256 * i + 255 < KCH) || (256 * i < KCH && g < KCH);
      const int c = isk ? g : g - KCH;
      const int row = c / (D / 8), ch = c % (D / 8);
      const size_t r = (size_t)min(j0 + row, seg_len - 1);
      const bf16_t* src = isk ? kbase + r * k_stride + ch * 8 : vbase + r * v_stride + ch * 8;
      rs[i] = *reinterpret_cast<const u32x4_t*>(src);
    }
  };
  auto lstore = [&](int buf) {
    bf16_t* Ks = Ks2[buf];
    bf16_t* Vs = Vs2[buf];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int g = tid + 256 * i;
      const bool isk = (256 * i + 255 < KCH) || (256 * i < KCH && g < KCH);
      const int c = isk ? g : g - KCH;
      const int row = c / (D / 8), ch = c % (D / 8);
      bf16_t* dst = isk ? &Ks[row * K_LD + ch * 8] : &Vs[row * V_LD + ch * 8];
      *reinterpret_cast<u32x4_t*>(dst) = rs[i];
    }
  };

  if (ONES) {
    constexpr int XC = (DP - D) / 8;
    for (int c = tid; c < 2 * BKV * XC; c += 256) {
      const int buf = c / (BKV * XC), rem = c % (BKV * XC), row = rem / XC, x = rem % XC;
      const u32x4_t v = {x == 0 ? 0x00003F80u : 0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4_t*>(&Vs2[buf][row * V_LD + D + 8 * x]) = v;
    }
  }
  f32x16_t ot[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  if (ntiles > 0) gload(0);
  // Q is loop-invariant but was produced by vector loads: hipcc's waitcnt insertion keeps "Q may still be in flight"
  // alive around the loop back-edge and puts COUNTED vmcnt waits in front of the QK^T MFMAs of EVERY iteration -
  // counted against the newest loads, i.e. the MFMAs wait for the next tile's global loads (the prefetch distance of
  // one tile was really zero; found in the .s).  Drain once here and re-define Q through an empty asm.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks]));
  if (ntiles > 0) lstore(0);
  __syncthreads();

  // L2 prefetch two tiles ahead: one 4-byte load per thread touches the cache lines of tile t+2's K and V rows
  // (row = tid / 4; lines at byte 0 / 128 of the K slice and of the V slice), so that the real 16-byte loads of the
  // next iteration find them in L2 instead of paying the HBM latency with a prefetch distance of a single tile
  // (register staging cannot run two tiles ahead: 28 more VGPRs do not exist here).  The value is only "used" to keep
  // the load alive.
  ST_DECL;
  ST_BEGIN();
  uint32_t pf_sink = 0;
  const int pf_row = tid >> 2;
  const bf16_t* pf_base = ((tid & 2) ? vbase : kbase) + (tid & 1) * 64;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
#ifndef ATTN_ABL_NO_GLOAD          // ablation builds of scripts/attn_probe.hip only (results are then meaningless)
    if (more) gload(t + 1);
#endif
    uint32_t pf_val = 0;
#ifndef ATTN_ABL_NO_PF
    // (D = 128 only - the long causal LLM prompt.  For the ViT's 576-key segments K / V of a head are L2-resident after the
    //  first query block and the extra load only costs issue slots: 55.4 -> 54.3 us per block without it)
    if (D == 128 && t + 2 < ntiles) {
      const int r = min((t + 2) * BKV + pf_row, seg_len - 1);
      pf_val = *reinterpret_cast<const uint32_t*>(pf_base + (size_t)r * ((tid & 2) ? v_stride : k_stride));
    }
#endif
    __builtin_amdgcn_sched_barrier(0);   // keep the loads up here (hipcc otherwise sinks the prefetch next to its use)
    ST_MARK(0);                          // global loads of tile t+1 issued
    const int j0 = t * BKV;
    const bf16_t* Ks = Ks2[t & 1];
    const bf16_t* Vs = Vs2[t & 1];

    // a wave whose 32 query rows all lie past the segment end (the last query block of a 576-patch image: rows 512..639)
    // only helps with the staging: its MFMA / VALU slots go to the other workgroup on the SIMD
    if (wave_rows) {
    // ---- S^T = K . Q^T for two 32-key blocks ----
    // (ks outer, kb inner: consecutive MFMAs go to different accumulators - a dependent 32x32x16 pair issues 64 cycles
    //  apart, an independent one 32; the accumulation order of each element is unchanged)
    f32x16_t st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
    const bf16_t* krow = &Ks[(lane & 31) * K_LD + 8 * h];
    if (D > 80) __builtin_amdgcn_s_setprio(1);      // (two waves per SIMD: measured 139.7 -> 128.6 us on a 4096-token causal prompt;
                                                    //  the three-wave instances lose 1-3 % with it, profiles/r05_attn_prefill_ab.txt)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(krow + kb * 32 * K_LD + ks * 16);
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kb], 0, 0, 0);
      }
    if (D > 80) __builtin_amdgcn_s_setprio(0);

    // ---- online softmax (q = lane&31 is lane-local; lane^32 holds the other keys) ----
    // The loop is VALU-bound (22 MFMAs per tile vs the element-wise work on 32 scores per lane), so the
    // element-wise part is kept to one fma + one v_exp_f32 + one add per score:
    //   * masking (segment end / causal diagonal) only on the tiles that need it (wave-uniform test);
    //   * the softmax scale is folded into the exponent: p = exp2(s * scale_log2 - m), the running max is
    //     taken on the raw scores;
    //   * "deferred max": the reference max m_run only moves when some row's tile max exceeds it by more
    //     than 2^8 (log2 domain) - P stays <= 256, exact in the normalised result, and the O / l rescale
    //     (48 accumulator registers) runs on a few tiles per query block instead of every tile.
    ST_MARK(1);                          // QK^T MFMAs issued
    const bool need_mask = (j0 + BKV > seg_len) || (CAUSAL && j0 + BKV - 1 > qbase + wave * 32);
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const bool ok = key < seg_len && (!CAUSAL || key <= qrow);
          st[kb][r] = ok ? st[kb][r] : -INFINITY;
        }
    }
    float mt = __builtin_elementwise_maximum(st[0][0], st[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = __builtin_elementwise_maximum(mt, __builtin_elementwise_maximum(st[0][r], st[1][r]));
    {
      float ma, mb;
      vlm_xor32_pair(mt, ma, mb);                                 // lanes l / l ^ 32 in one v_permlane32_swap (no LDS round trip)
      mt = fmaxf(ma, mb) * scale_log2;                            // scale_log2 > 0
    }
    if (__any(mt > m_run + 8.0f)) {                                // wave-uniform: move the reference max
      const float m_new = fmaxf(m_run, mt);
      const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_new);   // m_run = -inf -> 0
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] *= alpha;
    }
    const float neg_m = (m_run == -INFINITY) ? 0.f : -m_run;
    float ls = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[kb][r], scale_log2, neg_m));
        st[kb][r] = p;
        if (!ONES) ls += p;
      }
    if (!ONES) l_run += ls;

    // ---- P fragments (B operand of O^T): registers 8m..8m+7 of block kb ----
    bf16x8_t pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const u32x4_t u = {pack_bf2(st[kb][8 * mm + 0], st[kb][8 * mm + 1]), pack_bf2(st[kb][8 * mm + 2], st[kb][8 * mm + 3]),
                           pack_bf2(st[kb][8 * mm + 4], st[kb][8 * mm + 5]), pack_bf2(st[kb][8 * mm + 6], st[kb][8 * mm + 7])};
        pf[kb][mm] = __builtin_bit_cast(bf16x8_t, u);
      }
    bf16x8_t pl[2][2];
    if (HILO) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const u32x4_t hi = __builtin_bit_cast(u32x4_t, pf[kb][mm]);
          u32x4_t lo;
#pragma unroll
          for (int w = 0; w < 4; ++w)
            lo[w] = pack_bf2(st[kb][8 * mm + 2 * w] - bf_lo(hi[w]), st[kb][8 * mm + 2 * w + 1] - bf_hi(hi[w]));
          pl[kb][mm] = __builtin_bit_cast(bf16x8_t, lo);
        }
    }

    ST_MARK(2);                          // softmax + P pack (waits for the QK^T results)
    // ---- O^T += V^T . P^T ----
    // lane (i = lane & 15, g = lane >> 4): d = db * 32 + 16 (g & 1) + i, keys K0 + {0..3} and K0 + 8 + {0..3} with
    // K0 = kb * 32 + 16 mm + 4 h; it passes the address of row K0 + i / 4, columns (its group's 16 d) + 4 (i % 4)
    const bf16_t* vlane = &Vs[(4 * h + ((lane & 15) >> 2)) * V_LD + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)];
    if (D > 80) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int db = 0; db < NDB; ++db) {      // db innermost: NDB independent accumulator chains interleave
          const bf16_t* p0 = vlane + (kb * 32 + 16 * mm) * V_LD + db * 32;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + 8 * V_LD));
          const bf16x8_t vf = __builtin_shufflevector(__builtin_bit_cast(bf16x4_t, lo), __builtin_bit_cast(bf16x4_t, hi), 0, 1, 2, 3, 4,
                                                      5, 6, 7);
          ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][mm], ot[db], 0, 0, 0);
          if (HILO) ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pl[kb][mm], ot[db], 0, 0, 0);
        }
    if (D > 80) __builtin_amdgcn_s_setprio(0);
    }   // wave_rows

    // hipcc otherwise hoists the register-only part of lstore (the V^T pair packing) up between the QK^T MFMAs - with
    // the s_waitcnt vmcnt it needs: the MFMAs then wait for the NEXT tile's global loads (found in the .s)
    __builtin_amdgcn_sched_barrier(0);
    ST_MARK(3);                          // P.V MFMAs issued
#ifndef ATTN_ABL_NO_LSTORE
    if (more) lstore((t + 1) & 1);   // the other buffer: last read in iteration t-1, one barrier ago
#endif
    ST_MARK(4);                          // staging registers -> LDS (waits for tile t+1's global loads)
    __syncthreads();
    ST_MARK(5);                          // barrier
    pf_sink += pf_val;   // consumed only here, behind the staging loads' own wait: the prefetch never stalls anything
  }

  asm volatile("" ::"v"(pf_sink));
#ifdef ATTN_STAMPS
  if (st_on && lane == 0)
    for (int i = 0; i < 6; ++i) g_attn_stamps[wave][i] = st_sum[i];
#endif
  // ---- normalise and store: lane holds O^T[d = db*32 + (r&3) + 8(r>>2) + 4h][q = lane&31] ----
  float l_a, l_b;
  vlm_xor32_pair(ONES ? ot[LDB][LR] : l_run, l_a, l_b);
  const float l_tot = ONES ? l_a : l_a + l_b;
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (qrow < seg_len) {
    bf16_t* orow = op + (size_t)(seg_start + qrow) * o_stride + (size_t)head * D;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int d0 = db * 32 + 8 * r4 + 4 * h;
        if (d0 < D) {
          uint2 o;
          o.x = pack_bf2(ot[db][4 * r4 + 0] * inv, ot[db][4 * r4 + 1] * inv);
          o.y = pack_bf2(ot[db][4 * r4 + 2] * inv, ot[db][4 * r4 + 3] * inv);
          *reinterpret_cast<uint2*>(orow + d0) = o;
        }
      }
  }
}

}  // namespace

extern "C" int vlm_attn_prefill(const void* q, const void* k, const void* v, void* out, int q_stride, int k_stride,
                                int v_stride, int o_stride, const void* cu_seqlens, int nseg, int total_qblocks, int Hq,
                                int Hkv, int D, float scale, int causal, void* stream) {
  if (!q || !k || !v || !out || !cu_seqlens || nseg <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0) return VLM_ERR_ARG;
  if (q_stride % 8 || k_stride % 8 || v_stride % 8 || o_stride % 4) return VLM_ERR_SHAPE;
  if (total_qblocks <= 0) return VLM_OK;
  hipStream_t st = (hipStream_t)stream;
  const float sl2 = scale * 1.44269504088896340736f;
  dim3 grid(total_qblocks, Hq), block(256);
  // bit 1 of `causal`: the caller asserts that all segments have the same length -> XCD-local placement
  const bool uniform = (causal & 2) != 0 && total_qblocks % nseg == 0 && ((long)nseg * Hq) % 8 == 0;
  const int uniform_nqb = uniform ? total_qblocks / nseg : 0;
  // bit 2 (round 6): cu_seqlens is followed by q_start int32 [nseg] (see the header); total_qblocks counts the query rows only
  const int* qstart = (causal & 4) ? (const int*)cu_seqlens + nseg + 1 : nullptr;
  if ((causal & 4) && (causal & 2)) return VLM_ERR_ARG;
  causal &= 1;
  // P as hi + lo bf16 operands (16 mantissa bits into P.V; the reference's fused attention keeps P in fp32).  Measured on MI355X
  // (profiles/r06_attention_p_hilo.txt): distance to the exactly rounded result 1.8e-3 -> 7e-5; launch time +25 % (ViT, D = 80:
  // 52.5 -> 66 us at 16 images) / +28 % (LLM prompt, D = 128 causal, 4096 tokens: 130 -> 167 us).  Default: ON for D = 128 - the
  // language model, where the prompt's attention is a small share of the time to the first token and its rounding walks through 28
  // layers of logits - and OFF for the vision towers (D = 64 / 80), where the launch is 11 % of the images/s metric's block.
  // VLM_ATTN_PREFILL_HILO=0 / 1 forces every instance.
  static const int hilo_env = [] { const char* e = getenv("VLM_ATTN_PREFILL_HILO"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  const bool hilo = hilo_env >= 0 ? hilo_env == 1 : D == 128;
#define GO(DV, CV)                                                                                                    \
  do {                                                                                                                \
    if (hilo)                                                                                                         \
      hipLaunchKernelGGL((attn_prefill_kernel<DV, CV, true>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,  \
                         (const bf16_t*)v, (bf16_t*)out, q_stride, k_stride, v_stride, o_stride, (const int*)cu_seqlens, \
                         nseg, Hq, Hkv, sl2, uniform_nqb, qstart);                                                             \
    else                                                                                                              \
      hipLaunchKernelGGL((attn_prefill_kernel<DV, CV, false>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k, \
                         (const bf16_t*)v, (bf16_t*)out, q_stride, k_stride, v_stride, o_stride, (const int*)cu_seqlens, \
                         nseg, Hq, Hkv, sl2, uniform_nqb, qstart);                                                             \
  } while (0)
  if (D == 80 && !causal) GO(80, false);
  else if (D == 80 && causal) GO(80, true);
  else if (D == 128 && !causal) GO(128, false);
  else if (D == 128 && causal) GO(128, true);
  else if (D == 64 && !causal) GO(64, false);
  else if (D == 64 && causal) GO(64, true);
  else return VLM_ERR_SHAPE;
#undef GO
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
