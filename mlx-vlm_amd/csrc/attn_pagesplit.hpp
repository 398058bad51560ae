// The page walk of the page-split decode attention (attn_decode.hip) and the merge of its partials (gemv_bf16.hip) as
// shared device functions (the rejected fused decode block, scripts/rejected/decode_block.hip.txt, was built from the same
// sources to be bit-identical with the launches it replaced): ONE wave walks pages s, s + S, s + 2S, ... of (sequence b, kv head g) and leaves the unnormalised
// O^T[d = 16 dt + 4 gq + r][head = lane & 15] in ot, the running max (log2 domain) in m_run and this lane's share of the row
// sum in l_run.  Layouts, MFMA operand order and the reasons for them: attn_decode.hip (file header).
#pragma once
#include "common.hpp"

constexpr int VLM_HD = 128;    // head_dim supported by the decode path
constexpr int VLM_PAGE = 64;

// P as TWO bf16 MFMA operands: hi = bf16(p), lo = bf16(p - hi).  The reference's fused attention keeps P in fp32
// (mx.fast.scaled_dot_product_attention, reference base.py:366); a single bf16 P was the one rounding point of the engine that the
// typed graph does not have (profiles/r05_engine_noise_by_depth_and_op.txt: 1.8e-3 from the exactly rounded result where every
// GEMM is at 1e-5).  hi + lo carries 16 mantissa bits of p into O^T += V^T . P^T at the price of a second MFMA per fragment - free
// in the decode kernels, whose launch is one memory round trip long whatever the matrix pipe does.  st[t][r] = p(key 16 t + 4 gq + r);
// k-slot 8 gq + j of step u <- tile 2u (j < 4) / tile 2u + 1 (j >= 4), register j & 3.
__device__ __forceinline__ void vlm_pack_p_hilo(const f32x4_t (&st)[4], bf16x8_t (&ph)[2], bf16x8_t (&pl)[2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    u32x4_t hi, lo;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float a = st[2 * u + (w >> 1)][2 * (w & 1)], b = st[2 * u + (w >> 1)][2 * (w & 1) + 1];
      const uint32_t h = pack_bf2(a, b);
      hi[w] = h;
      lo[w] = pack_bf2(a - bf_lo(h), b - bf_hi(h));
    }
    ph[u] = __builtin_bit_cast(bf16x8_t, hi);
    pl[u] = __builtin_bit_cast(bf16x8_t, lo);
  }
}

template <int G, bool IDENT>
__device__ __forceinline__ void vlm_pagesplit_walk(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kpool,
                                                   const bf16_t* __restrict__ vpool, const int* __restrict__ block_table,
                                                   const int* __restrict__ kv_len, int ldq, int max_pages, int Hkv, int kv_len_add,
                                                   float scale_log2, int S, int b, int g, int s, int lane, f32x4_t (&ot)[8],
                                                   float& m_run, float& l_run, int& npages) {
  constexpr int HD = VLM_HD, PAGE = VLM_PAGE;
  const int head = lane & 15, gq = lane >> 4;
  int pi = s;
  const int* trow = IDENT ? nullptr : block_table + (size_t)b * max_pages;
  size_t page = IDENT ? (size_t)b * max_pages + min(pi, max_pages - 1) : (size_t)trow[min(pi, max_pages - 1)];
  int len_raw;
  asm volatile("global_load_dword %0, %1, off" : "=v"(len_raw) : "v"(kv_len + b) : "memory");
  bf16x8_t qf[4];
  {
    const bf16_t* qr = q + (size_t)b * ldq + (size_t)(g * G + min(head, G - 1)) * HD + 8 * gq;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[ds] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(qr + 32 * ds));
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) ot[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  m_run = -INFINITY;
  l_run = 0.f;
  int len = 0;
  npages = 0;
  do {
    if (!IDENT) asm volatile("s_waitcnt vmcnt(0)" : "+v"(len_raw), "+v"(page)::"memory");
    const bf16_t* kp = kpool + (page * Hkv + g) * (size_t)(HD / 8) * PAGE * 8 + ((size_t)gq * PAGE + head) * 8;
    const bf16_t* vp = vpool + (page * Hkv + g) * (size_t)HD * PAGE + (size_t)head * PAGE + 8 * gq;
    u32x4_t kf[4][4], vf[8][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ds = 0; ds < 4; ++ds)
        kf[t][ds] = *reinterpret_cast<const u32x4_t*>(kp + ((size_t)(4 * ds) * PAGE + 16 * t) * 8);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        vf[dt][u] = *reinterpret_cast<const u32x4_t*>(vp + (size_t)(16 * dt) * PAGE + 32 * u);
    const int pc = pi;     // the page being processed
    pi += S;
    const size_t next_page = IDENT ? (size_t)b * max_pages + min(pi, max_pages - 1) : (size_t)trow[min(pi, max_pages - 1)];
    __builtin_amdgcn_sched_barrier(0);
    if (!IDENT) {
      len = __builtin_amdgcn_readfirstlane(len_raw) + kv_len_add;
      npages = (len + PAGE - 1) / PAGE;
    }
    f32x4_t st[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      st[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < 4; ++ds)
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[t][ds]), qf[ds], st[t], 0, 0, 0);
    }
    if (IDENT) {
      asm volatile("s_waitcnt vmcnt(16)" : "+v"(len_raw)::"memory");
      len = __builtin_amdgcn_readfirstlane(len_raw) + kv_len_add;
      npages = (len + PAGE - 1) / PAGE;
    }
    float mt = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = pc * PAGE + 16 * t + 4 * gq + r;
        const float sv = key < len ? st[t][r] * scale_log2 : -INFINITY;
        st[t][r] = sv;
        mt = fmaxf(mt, sv);
      }
    mt = col4_max(mt);
    const float m_new = fmaxf(m_run, mt);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_use);
    float ls = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f(st[t][r] - m_use);
        st[t][r] = p;
        ls += p;
      }
    l_run = l_run * alpha + ls;
    m_run = m_new;
    bf16x8_t pb[2], pl[2];
    vlm_pack_p_hilo(st, pb, pl);
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4_t vv = vf[dt][u];
        const int k0 = pc * PAGE + 32 * u + 4 * gq, k1 = k0 + 16;
        vv[0] = (k0 + 1 < len) ? vv[0] : ((k0 < len) ? (vv[0] & 0xffffu) : 0u);
        vv[1] = (k0 + 3 < len) ? vv[1] : ((k0 + 2 < len) ? (vv[1] & 0xffffu) : 0u);
        vv[2] = (k1 + 1 < len) ? vv[2] : ((k1 < len) ? (vv[2] & 0xffffu) : 0u);
        vv[3] = (k1 + 3 < len) ? vv[3] : ((k1 + 2 < len) ? (vv[3] & 0xffffu) : 0u);
        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pb[u], ot[dt], 0, 0, 0);
        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pl[u], ot[dt], 0, 0, 0);
      }
    }
    if (pi >= npages) break;
    page = next_page;
  } while (true);

}

// Merge of the page-split partials for ONE 8-element chunk of the attention output (the tail of
// mx.fast.scaled_dot_product_attention, reference base.py:366-373):
//     x[d] = sum_s f_s O_s[d] / sum_s f_s l_s,   f_s = 2^(m_s - M)
// ml[sp] = (m, l) of split sp (m in the log2 domain of the walk above, -inf = the split owns no page and its O bytes may be
// anything), o[sp] = its 8 fp32 O values (round 6: fp32 - bf16 partials were a second rounding point the reference's fused
// attention does not have, 2.4e-3 from the exactly rounded result with everything else at 1e-5: tests/test_op_noise_gpu.py);
// every split up to NS is passed, the ones >= S are dropped by a select.  Used by the o_proj prologue (gemv_bf16.hip, PRO_ATTN_PS).
constexpr int VLM_MERGE_S = 16;
template <int NS>
__device__ __forceinline__ uint4 vlm_merge_splits16(const float2 (&ml)[NS], const f32x4_t (&o)[NS][2], int S) {
  float mm = -INFINITY;
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) mm = fmaxf(mm, sp < S ? ml[sp].x : -INFINITY);
  float ll = 0.f, acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sp = 0; sp < NS; ++sp) {
    const bool on = sp < S && ml[sp].x != -INFINITY;
    const float f = on ? exp2f(ml[sp].x - mm) : 0.f;
    ll += on ? f * ml[sp].y : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc8[j] += on ? f * o[sp][j >> 2][j & 3] : 0.f;
  }
  const float il = 1.0f / ll;
  uint4 r;
  r.x = pack_bf2(acc8[0] * il, acc8[1] * il);
  r.y = pack_bf2(acc8[2] * il, acc8[3] * il);
  r.z = pack_bf2(acc8[4] * il, acc8[5] * il);
  r.w = pack_bf2(acc8[6] * il, acc8[7] * il);
  return r;
}
