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
// 16m+8+4h+{0..3} (h = lane>>5), and V^T is read from LDS with exactly that key
// permutation (two ds_read_b64 per fragment).  K rows are padded by 16 B and the
// V^T rows by 8 B so the 16-lane ds_read_b128 / 32-lane ds_read_b64 groups are
// bank-conflict free.  The next K/V tile's global loads are issued before the
// MFMAs of the current one (register staging, T14-style) and written to LDS after
// the barrier.  Output is stored as 8-byte bf16x4 pieces (4 consecutive d).
#include "common.cuh"
#include "../../include/vlm_hip.h"

namespace {

constexpr int BQ = 128;   // query rows per workgroup (4 waves x 32)
constexpr int BKV = 64;   // keys per tile
constexpr int VT_LD = BKV + 4;

template <int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_prefill_kernel(
    const bf16_t* __restrict__ qp, const bf16_t* __restrict__ kp, const bf16_t* __restrict__ vp, bf16_t* __restrict__ op,
    int q_stride, int k_stride, int v_stride, int o_stride, const int* __restrict__ cu, int nseg, int Hq, int Hkv,
    float scale_log2, int uniform_nqb) {
  constexpr int DP = (D + 31) / 32 * 32;   // padded head dim for the O^T tiles
  constexpr int NKS = D / 16;              // k-steps of the S^T product
  constexpr int NDB = DP / 32;             // 32-wide d blocks of O^T
  constexpr int K_LD = D + 8;              // padded K row (elements)
  constexpr int KCH = BKV * D / 8;         // 16-byte chunks in a K tile
  constexpr int K_PER = (KCH + 255) / 256;
  constexpr int VIT = (BKV / 2) * (D / 8); // (key pair, d-chunk) items in a V tile
  constexpr int V_PER = (VIT + 255) / 256;

  // double-buffered: tile t+1 is written while the other waves may still read tile t -> ONE barrier per tile
  __shared__ __attribute__((aligned(16))) bf16_t Ks2[2][BKV * K_LD];
  __shared__ __attribute__((aligned(16))) bf16_t Vt2[2][DP * VT_LD];

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
    int bid = blockIdx.x;
    for (; seg < nseg; ++seg) {
      const int nb = (cu[seg + 1] - cu[seg] + BQ - 1) / BQ;
      if (bid < nb) { qb = bid; break; }
      bid -= nb;
    }
    if (seg >= nseg) return;
  }
  const int seg_start = cu[seg], seg_len = cu[seg + 1] - seg_start;
  const int kvh = head / (Hq / Hkv);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  __builtin_assume(tid >= 0 && tid < 256);
  const int qrow = qb * BQ + wave * 32 + (lane & 31);          // row inside the segment
  const int qrow_c = min(qrow, seg_len - 1);

  // ---- Q fragments (B operand of S^T): q = lane&31, d = ks*16 + 8h .. +8 ----
  bf16x8_t qf[NKS];
  {
    const bf16_t* qr = qp + (size_t)(seg_start + qrow_c) * q_stride + (size_t)head * D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qr + ks * 16);
  }

  const int kv_end = CAUSAL ? min(seg_len, qb * BQ + BQ) : seg_len;
  const int ntiles = (kv_end + BKV - 1) / BKV;
  const bf16_t* kbase = kp + (size_t)seg_start * k_stride + (size_t)kvh * D;
  const bf16_t* vbase = vp + (size_t)seg_start * v_stride + (size_t)kvh * D;

  u32x4_t rk[K_PER], rv0[V_PER], rv1[V_PER];
  // (loads are unconditional - surplus lanes re-read the last chunk - so the staging
  //  registers are fully defined and stay in VGPRs; only the LDS store is guarded)
  auto gload = [&](int t) {
    const int j0 = t * BKV;
#pragma unroll
    for (int i = 0; i < K_PER; ++i) {
      const int c = min(tid + 256 * i, KCH - 1);
      const int row = c / (D / 8), ch = c % (D / 8);
      rk[i] = *reinterpret_cast<const u32x4_t*>(kbase + (size_t)min(j0 + row, seg_len - 1) * k_stride + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < V_PER; ++i) {
      const int c = min(tid + 256 * i, VIT - 1);
      const int kpair = c % (BKV / 2), ch = c / (BKV / 2);
      const bf16_t* v0 = vbase + (size_t)min(j0 + 2 * kpair, seg_len - 1) * v_stride + ch * 8;
      const bf16_t* v1 = vbase + (size_t)min(j0 + 2 * kpair + 1, seg_len - 1) * v_stride + ch * 8;
      rv0[i] = *reinterpret_cast<const u32x4_t*>(v0);
      rv1[i] = *reinterpret_cast<const u32x4_t*>(v1);
    }
  };
  auto lstore = [&](int buf) {
    bf16_t* Ks = Ks2[buf];
    bf16_t* Vt = Vt2[buf];
#pragma unroll
    for (int i = 0; i < K_PER; ++i) {
      const int c = tid + 256 * i;
      if ((KCH % 256 == 0) || c < KCH) {
        const int row = c / (D / 8), ch = c % (D / 8);
        *reinterpret_cast<u32x4_t*>(&Ks[row * K_LD + ch * 8]) = rk[i];
      }
    }
#pragma unroll
    for (int i = 0; i < V_PER; ++i) {
      const int c = tid + 256 * i;
      if ((VIT % 256 == 0) || c < VIT) {
        const int kpair = c % (BKV / 2), ch = c / (BKV / 2);
        // element 2e (low halves) and 2e+1 (high halves) of the two keys
#define VT_PUT(E, AW, BW)                                                                                         \
  *reinterpret_cast<uint32_t*>(&Vt[(ch * 8 + 2 * (E)) * VT_LD + 2 * kpair]) = ((AW) & 0xffffu) | ((BW) << 16);     \
  *reinterpret_cast<uint32_t*>(&Vt[(ch * 8 + 2 * (E) + 1) * VT_LD + 2 * kpair]) = ((AW) >> 16) | ((BW) & 0xffff0000u);
        VT_PUT(0, rv0[i].x, rv1[i].x)
        VT_PUT(1, rv0[i].y, rv1[i].y)
        VT_PUT(2, rv0[i].z, rv1[i].z)
        VT_PUT(3, rv0[i].w, rv1[i].w)
#undef VT_PUT
      }
    }
  };

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
  uint32_t pf_sink = 0;
  const int pf_row = tid >> 2;
  const bf16_t* pf_base = ((tid & 2) ? vbase : kbase) + (tid & 1) * 64;
  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) gload(t + 1);
    uint32_t pf_val = 0;
    if (t + 2 < ntiles) {
      const int r = min((t + 2) * BKV + pf_row, seg_len - 1);
      pf_val = *reinterpret_cast<const uint32_t*>(pf_base + (size_t)r * ((tid & 2) ? v_stride : k_stride));
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the loads up here (hipcc otherwise sinks the prefetch next to its use)
    const int j0 = t * BKV;
    const bf16_t* Ks = Ks2[t & 1];
    const bf16_t* Vt = Vt2[t & 1];

    // ---- S^T = K . Q^T for two 32-key blocks ----
    f32x16_t st[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
      const bf16_t* krow = &Ks[(kb * 32 + (lane & 31)) * K_LD + 8 * h];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(krow + ks * 16);
        st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kb], 0, 0, 0);
      }
    }

    // ---- online softmax (q = lane&31 is lane-local; lane^32 holds the other keys) ----
    // The loop is VALU-bound (22 MFMAs per tile vs the element-wise work on 32 scores per lane), so the
    // element-wise part is kept to one fma + one v_exp_f32 + one add per score:
    //   * masking (segment end / causal diagonal) only on the tiles that need it (wave-uniform test);
    //   * the softmax scale is folded into the exponent: p = exp2(s * scale_log2 - m), the running max is
    //     taken on the raw scores;
    //   * "deferred max": the reference max m_run only moves when some row's tile max exceeds it by more
    //     than 2^8 (log2 domain) - P stays <= 256, exact in the normalised result, and the O / l rescale
    //     (48 accumulator registers) runs on a few tiles per query block instead of every tile.
    const bool need_mask = (j0 + BKV > seg_len) || (CAUSAL && j0 + BKV - 1 > qb * BQ + wave * 32);
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
    float mt = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, st[kb][r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * scale_log2;         // scale_log2 > 0
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
        ls += p;
      }
    l_run += ls;

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

    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      const bf16_t* vrow = &Vt[(db * 32 + (lane & 31)) * VT_LD + 4 * h];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const uint2 lo = *reinterpret_cast<const uint2*>(vrow + kb * 32 + 16 * mm);
          const uint2 hi = *reinterpret_cast<const uint2*>(vrow + kb * 32 + 16 * mm + 8);
          const u32x4_t u = {lo.x, lo.y, hi.x, hi.y};
          const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, u);
          ot[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb][mm], ot[db], 0, 0, 0);
        }
    }

    // hipcc otherwise hoists the register-only part of lstore (the V^T pair packing) up between the QK^T MFMAs - with
    // the s_waitcnt vmcnt it needs: the MFMAs then wait for the NEXT tile's global loads (found in the .s)
    __builtin_amdgcn_sched_barrier(0);
    if (more) lstore((t + 1) & 1);   // the other buffer: last read in iteration t-1, one barrier ago
    __syncthreads();
    pf_sink += pf_val;   // consumed only here, behind the staging loads' own wait: the prefetch never stalls anything
  }

  asm volatile("" ::"v"(pf_sink));
  // ---- normalise and store: lane holds O^T[d = db*32 + (r&3) + 8(r>>2) + 4h][q = lane&31] ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
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
  causal &= 1;
#define GO(DV, CV)                                                                                                    \
  hipLaunchKernelGGL((attn_prefill_kernel<DV, CV>), grid, block, 0, st, (const bf16_t*)q, (const bf16_t*)k,            \
                     (const bf16_t*)v, (bf16_t*)out, q_stride, k_stride, v_stride, o_stride, (const int*)cu_seqlens,  \
                     nseg, Hq, Hkv, sl2, uniform_nqb)
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
