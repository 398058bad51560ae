// Per-step decode attention over the paged KV cache for gfx950, on the matrix cores.
//
// Replaces mx.fast.scaled_dot_product_attention for L == 1
//   (reference mlx_vlm/models/base.py:366-373 called from
//    mlx_vlm/models/qwen2_vl/language.py:115-118; GQA n_heads / n_kv_heads from
//    language.py:44-49), mask=None (base.py:214-228: N == 1 -> no mask).
//
// The op is tiny (0.5 MB of K/V per layer at ctx 512) and purely latency bound, so the design minimises
// the length of the dependent instruction chain, not bytes:
//   * the GQA group's G query heads (6 for Qwen2-VL-2B) are the N dimension of a 16x16x32 MFMA, one
//     64-key page is 4 key tiles: S^T[key][head] = K . Q^T takes 16 MFMAs per page and O^T[d][head] =
//     V^T . P^T another 16 - ~600 cycles instead of ~6000 cycles of VALU dot products and shuffles;
//   * both products are "transposed" so the C/D column is the query head: every softmax statistic is
//     lane-local (head = lane & 15), the only cross-lane traffic per page is two xor-shuffles (16, 32) for
//     the max, and the exponentiated S^T registers ARE the B operand of the second MFMA (registers of key
//     tiles 2u and 2u+1 -> the 8 k-slots of 32-key step u);
//   * the KV pools are laid out so that every MFMA operand fragment is ONE 16-byte global load straight
//     into VGPRs (no LDS staging): K pool [page][Hkv][D/8][64 keys][8]  (A fragment: key = lane&15, 8 d),
//     V pool [page][Hkv][D][64 key slots] with the keys of each 32-block stored in k-slot order
//     (slot = 8*((k&15)>>2) + 4*(k>>4) + (k&3)), so the V^T fragment (d = lane&15, 8 key slots) is
//     contiguous too.  All 32 loads of a page are issued before the first MFMA.
//   * one wave per page, 16 waves per workgroup, merged through LDS.  nsplit == 1: the workgroup owns the
//     whole (sequence, kv head) and writes the final bf16 vector; nsplit > 1: workgroups write (m, l, O)
//     partials that the o_proj GEMV prologue (vlm_gemv_attn_out) or the combine kernel merges.
// fp32 scores / statistics / accumulation; P is rounded to bf16 for the second MFMA (as in the prefill
// flash kernel; tolerance stated in tests).
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "attn_pagesplit.hpp"
#include "internal.h"
#include "../../include/vlm_hip.h"

namespace {

constexpr int HD = VLM_HD;   // head_dim supported by the decode path
constexpr int PAGE = VLM_PAGE;

// NW waves per workgroup: 16 (one round covers 1024 tokens of context); 8 when G == 8 (LDS merge buffer <= 64 KB)
template <int G, int NW, bool STAMPS, bool IDENT>
__global__ __launch_bounds__(NW * 64) void attn_decode_mfma_kernel(
    // the first 14 dwords are everything the first loads need: with -amdgpu-kernarg-preload-count=16 they are in
    // SGPRs at wave start (no scalar-load round trip before the table / length / Q loads)
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ kpool, const bf16_t* __restrict__ vpool,
    const int* __restrict__ block_table, const int* __restrict__ kv_len, int ldq, int max_pages, int Hkv, int kv_len_add,
    int Hq, float scale_log2, int nsplit, int ldo, float* __restrict__ part_o, float* __restrict__ part_ml,
    bf16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float red_o[NW][G][HD];
  __shared__ float red_m[NW][G], red_l[NW][G];

  const int b = blockIdx.x / Hkv, g = blockIdx.x % Hkv, s = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int head = lane & 15, gq = lane >> 4;
  // debug timeline (VLM_ATTN_STAMPS=1): 100 MHz wall clock stamps of wave 0 / 15 of workgroup 0 -> part_o
  unsigned long long t0 = 0;
  auto stamp = [&](int i) {
    if (STAMPS && blockIdx.x == 0 && (wave == 0 || wave == NW - 1) && lane == 0) {
      const unsigned long long t = wall_clock64();
      if (i == 0) t0 = t;
      part_o[(wave == 0 ? 0 : 16) + i] = (float)(t - t0) * 0.01f;   // microseconds since the wave started
    }
  };
  stamp(0);
  // Latency plan (the kernel is a chain of dependent first-touch loads, 1-2 us each, so the chain is kept at
  // three links: kernel arguments -> {table entry, context length, Q} -> K/V):
  //   * every kernel argument is pulled into SGPRs in ONE scalar-load batch at entry (the empty asm "uses" them;
  //     otherwise the compiler fetches them lazily in three separate round trips);
  //   * the context length is read through the vector memory path (inline asm; the explicit vmcnt(0) at the top
  //     of the page loop covers it): as a scalar load its wait would also block the kernel-argument loads
  //     (scalar loads return out of order, lgkmcnt(0) is the only wait), a volatile load is waited for at once;
  //   * the first page's table entry is the FIRST vector load, Q and the length follow; vector loads return in
  //     order, so the K/V address computation waits for the table entry only;
  //   * heads >= G of the 16-wide MFMA N dimension read head G-1 again instead of a select-to-zero on loaded
  //     data (their columns are never written), so nothing waits on Q before the first MFMA;
  //   * every wave processes its first page UNCONDITIONALLY (do-while): a wave whose page lies beyond the context
  //     computes on masked scores (m = -inf, l = 0, O = 0) and drops out in the merge.  Block-table rows are
  //     zero-initialised and the pools fully mapped, so its loads are harmless.
  // IDENT (block_table == NULL): sequence b owns pages [b * max_pages, (b + 1) * max_pages) of the pools as passed -
  // no table load at all, the K/V loads are the first loads of the kernel
  int pi = s * NW + wave;
  const int* trow = IDENT ? nullptr : block_table + (size_t)b * max_pages;
  size_t page = IDENT ? (size_t)b * max_pages + min(pi, max_pages - 1) : (size_t)trow[min(pi, max_pages - 1)];
  int len_raw;
  asm volatile("global_load_dword %0, %1, off" : "=v"(len_raw) : "v"(kv_len + b) : "memory");

  // Q fragments (B operand of S^T): head = lane&15, d = 32*ds + 8*gq .. +8
  bf16x8_t qf[4];
  {
    const bf16_t* qr = q + (size_t)b * ldq + (size_t)(g * G + min(head, G - 1)) * HD + 8 * gq;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[ds] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(qr + 32 * ds));
  }
  __builtin_amdgcn_sched_barrier(0);

  f32x4_t ot[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ot[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  stamp(1);

  int len = 0, npages = 0;
  do {
    // table entry (+ context length and Q on the first pass) have landed; later passes: next_page, long since.
    // IDENT: nothing to wait for - the page number is arithmetic, K/V go out right behind the length / Q loads
    if (!IDENT) asm volatile("s_waitcnt vmcnt(0)" : "+v"(len_raw), "+v"(page)::"memory");
    const bf16_t* kp = kpool + (page * Hkv + g) * (size_t)(HD / 8) * PAGE * 8 + ((size_t)gq * PAGE + head) * 8;
    const bf16_t* vp = vpool + (page * Hkv + g) * (size_t)HD * PAGE + (size_t)head * PAGE + 8 * gq;
    // ---- every operand fragment of the page: 16 + 16 loads of 16 B, all in flight
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
    // the next page's table entry rides behind them (clamped index: unconditional, no dependent wait later)
    pi += nsplit * NW;
    const size_t next_page = IDENT ? (size_t)b * max_pages + min(pi, max_pages - 1) : (size_t)trow[min(pi, max_pages - 1)];
    __builtin_amdgcn_sched_barrier(0);   // all 32 loads are in flight before anything waits on one of them
    const int pc = pi - nsplit * NW;     // the page being processed
    if (!IDENT) {
      len = __builtin_amdgcn_readfirstlane(len_raw) + kv_len_add;
      npages = (len + PAGE - 1) / PAGE;
    }

    stamp(2);
    // ---- S^T = K . Q^T : st[t][r] = score(key = 16t + 4gq + r, head = lane&15)
    f32x4_t st[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      st[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < 4; ++ds)
        st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[t][ds]), qf[ds], st[t], 0, 0, 0);
    }
    if (STAMPS) { asm volatile("" :: "v"(st[3][3])); stamp(3); }
    if (IDENT) {
      // the length load is older than every K load (vector loads return in order) and the MFMAs above have
      // consumed K: at most the 16 V loads are still in flight
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
    const float m_use = m_new == -INFINITY ? 0.f : m_new;    // a wave with no valid key yet: p = exp2(-inf - 0) = 0
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
    l_run = l_run * alpha + ls;     // per-lane partial (keys of this gq); lanes of one head are summed at the end
    m_run = m_new;
    stamp(4);
    // ---- P^T fragments: k-slot 8gq + j of step u  <-  tile 2u (j < 4) / tile 2u+1 (j >= 4), register j & 3
    bf16x8_t pb[2], pl[2];          // hi + lo: 16 mantissa bits of p (attn_pagesplit.hpp)
    vlm_pack_p_hilo(st, pb, pl);
    // ---- O^T += V^T . P^T ; never-written V slots are multiplied by p == 0 but may hold NaN patterns -> zeroed
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        u32x4_t vv = vf[dt][u];
        // slots 8gq + j: keys 32u + 4gq + j (j<4) and 32u + 16 + 4gq + (j-4); mask whole words by validity
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

  if (STAMPS) { asm volatile("" :: "v"(ot[7][3])); stamp(5); }
  // ---- merge the waves of the workgroup: ot[dt][r] = O^T[d = 16dt + 4gq + r][head]
  l_run = col4_sum(l_run);
  if (head < G) {
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
      *reinterpret_cast<float4*>(&red_o[wave][head][16 * dt + 4 * gq]) = make_float4(ot[dt][0], ot[dt][1], ot[dt][2], ot[dt][3]);
    if (gq == 0) { red_m[wave][head] = m_run; red_l[wave][head] = l_run; }
  }
  stamp(6);
  __syncthreads();
  stamp(7);
  for (int i = tid; i < G * HD; i += NW * 64) {
    const int gg = i / HD, d = i % HD;
    float mm = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) mm = fmaxf(mm, red_m[w][gg]);
    float acc = 0.f, ll = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float f = red_m[w][gg] == -INFINITY ? 0.f : exp2f(red_m[w][gg] - mm);
      acc += f * red_o[w][gg][d];
      ll += f * red_l[w][gg];
    }
    if (out) {
      out[(size_t)b * ldo + (size_t)(g * G + gg) * HD + d] = f2bf(acc / ll);
    } else {
      // partials in the natural-log domain the merge code expects: m_e = m_2 * ln2
      const size_t hidx = ((size_t)b * Hq + (g * G + gg)) * nsplit + s;
      part_o[hidx * HD + d] = acc;
      if (d == 0) { part_ml[hidx * 2] = mm * 0.69314718055994530942f; part_ml[hidx * 2 + 1] = ll; }
    }
  }
  stamp(8);
  // pacing word of the weight prefetcher (csrc/prefetch.hip): a hint, nothing is ordered by it
}

// ------------------------------------------------------------------------------------------------------------------
// Page-split form: ONE wave (= one workgroup) per (sequence, kv head, page stride), S workgroups per (sequence, kv head),
// the LAST workgroup to arrive merges the partials and writes the final bf16 vector - no second launch, no merge in the
// o_proj prologue.
//
// Why (round-2 profile, profiles/r02_attn_decode_timeline.txt): the one-workgroup-per-kv-head form pulls a kv head's
// whole context (0.3 MB at 600 tokens) through ONE compute unit's vector-memory path - "kv-issued" grew with the bytes,
// 2.9 us at ctx 300, 5.8 us at ctx 600 - and ran on 2 of 256 CUs.  Here a page's 32 KB is the whole load of a CU, so
// the K/V arrival time is one memory round trip whatever the context, and the pages of a context are spread over
// S * Hkv CUs.  A workgroup walks pages s, s + S, s + 2S, ... (interleaved, so a short context still uses every
// workgroup that has a page).
// Hand-off (cdna_hip_programming.md Guideline 16, R1 form): partials (O^T registers -> [bh][s][head][d] fp32, m / l)
// leave with write-through (sc1) stores, every storing wave drains them (s_waitcnt vmcnt(0)), then ONE returning
// agent-scope atomic takes a ticket; the workgroup that draws S - 1 reads the partials back with sc1 loads (the
// producers stored sc1: no acquire fence needed), merges in fp32, stores bf16 and re-arms the ticket word (every
// workgroup has arrived by then).  Workgroups whose first page lies beyond the context store nothing and only arrive.
// tickets: one zero-initialised word per (sequence, kv head), owned by the caller, zero again after every launch.

// The tail of a page-split workgroup (shared by the bf16 and the 8-bit-KV kernels): ot[dt][r] = unnormalised O^T[d = 16 dt +
// 4 gq + r][head = lane & 15], m_run the running max (log2 domain), l_run this lane's share of the row sum.
template <int G, bool MERGE>
__device__ __forceinline__ void pagesplit_finish(f32x4_t (&ot)[8], float m_run, float l_run, int npages, int bh, int b, int g,
                                                 int s, int S, int Hkv, int lane, int ldo, float* part_o, float* part_ml,
                                                 unsigned* tickets, bf16_t* __restrict__ out) {
  const int head = lane & 15, gq = lane >> 4;
  l_run = col4_sum(l_run);
  if (!MERGE) {
    // partials for the o_proj prologue (vlm_gemv_attn_out_bf16): EVERY split writes (m, l) - a split with no page writes
    // (-inf, 0) and is skipped there - and the splits with pages their O^T as fp32 [b][head][s][d] (plain stores: the
    // kernel boundary publishes them; fp32 since round 6 - a bf16 partial is a rounding point the reference does not have).  m stays in the log2 domain of this kernel.
    if (head < G) {
      const size_t e = ((size_t)b * (Hkv * G) + (g * G + head)) * S + s;
      if (gq == 0) *reinterpret_cast<float2*>(part_ml + e * 2) = make_float2(m_run, l_run);
      if (m_run != -INFINITY) {
        float* po = part_o + e * HD + 4 * gq;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4_t*>(po + 16 * dt) = ot[dt];
      }
    }
    return;
  }
  const int n_act = min(S, npages);                // splits that own at least one page of this context
  const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(part_o, 0, 0x7fffffff, 0x00020000);   // (offsets checked by the launcher)
  typedef unsigned long long u64;
  if (s < n_act && head < G) {
    const int o0 = (((bh * S + s) * G + head) * HD + 4 * gq) * 4;                       // byte offset of d = 4 gq
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ot[dt]), rs_o, o0 + 64 * dt, 0, 16);   // aux 16 = sc1
    if (gq == 0)
      __hip_atomic_store(reinterpret_cast<u64*>(part_ml) + ((size_t)bh * S + s) * G + head,
                         ((u64)__float_as_uint(l_run) << 32) | __float_as_uint(m_run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                      // every storing wave drains (R1)
  unsigned tk = 0;
  if (lane == 0) tk = __hip_atomic_fetch_add(tickets + bh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  tk = __builtin_amdgcn_readfirstlane(tk);
  if (tk != (unsigned)S - 1u) return;

  // ---- last arriver: out[head][d] = sum_s f_s O_s[head][d] / sum_s f_s l_s,  f_s = 2^(m_s - M)
  constexpr int NJ = (G + 1) / 2;                  // float4 items per lane: G * 32 items over 64 lanes
  constexpr int CH = 8;                            // splits per pass (all loads of a pass in flight together)
  float M[NJ], L[NJ];
  f32x4_t acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { M[j] = -INFINITY; L[j] = 0.f; acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  for (int c0 = 0; c0 < n_act; c0 += CH) {
    u64 ml[CH][NJ];
    u32x4_t o[CH][NJ];
#pragma unroll
    for (int sp = 0; sp < CH; ++sp) {
      const int spc = min(c0 + sp, n_act - 1);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int item = min(lane + 64 * j, G * 32 - 1);
        ml[sp][j] = __hip_atomic_load(reinterpret_cast<const u64*>(part_ml) + ((size_t)bh * S + spc) * G + (item >> 5),
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o[sp][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_o, ((bh * S + spc) * G * HD + item * 4) * 4, 0, 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);             // every load of the pass is issued before the first use waits
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float mc = -INFINITY;
#pragma unroll
      for (int sp = 0; sp < CH; ++sp) mc = fmaxf(mc, (c0 + sp < n_act) ? __uint_as_float((unsigned)ml[sp][j]) : -INFINITY);
      const float mn = fmaxf(M[j], mc);             // finite: every split < n_act holds at least one valid key
      const float a = exp2f(M[j] - mn);
      L[j] *= a;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[j][r] *= a;
#pragma unroll
      for (int sp = 0; sp < CH; ++sp) {
        const float f = (c0 + sp < n_act) ? exp2f(__uint_as_float((unsigned)ml[sp][j]) - mn) : 0.f;
        L[j] += f * __uint_as_float((unsigned)(ml[sp][j] >> 32));
        const f32x4_t ov = __builtin_bit_cast(f32x4_t, o[sp][j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] += f * ov[r];
      }
      M[j] = mn;
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int item = lane + 64 * j;
    if (item < G * 32) {
      const float il = 1.0f / L[j];
      uint2 w;
      w.x = pack_bf2(acc[j][0] * il, acc[j][1] * il);
      w.y = pack_bf2(acc[j][2] * il, acc[j][3] * il);
      *reinterpret_cast<uint2*>(out + (size_t)b * ldo + (size_t)(g * G + (item >> 5)) * HD + 4 * (item & 31)) = w;
    }
  }
  if (lane == 0) __hip_atomic_store(tickets + bh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
}

template <int G, bool IDENT, bool MERGE>
__global__ __launch_bounds__(64) void attn_decode_pagesplit_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ kpool, const bf16_t* __restrict__ vpool,
    const int* __restrict__ block_table, const int* __restrict__ kv_len, int ldq, int max_pages, int Hkv, int kv_len_add,
    float scale_log2, int S, int ldo, float* part_o, float* part_ml, unsigned* tickets, bf16_t* __restrict__ out,
    int unused_) {
  const int bh = blockIdx.x, b = bh / Hkv, g = bh % Hkv, s = blockIdx.y;
  const int lane = threadIdx.x, head = lane & 15, gq = lane >> 4;
  // every kernel argument in ONE scalar-load batch at entry (the tail's arguments would otherwise be fetched lazily,
  // one more round trip on the path to the ticket)
  asm volatile("" ::"s"(part_o), "s"(part_ml), "s"(tickets), "s"(out), "s"(S), "s"(ldo));
  f32x4_t ot[8];
  float m_run, l_run;
  int npages;
  vlm_pagesplit_walk<G, IDENT>(q, kpool, vpool, block_table, kv_len, ldq, max_pages, Hkv, kv_len_add, scale_log2, S, b, g, s, lane, ot,
                               m_run, l_run, npages);

  pagesplit_finish<G, MERGE>(ot, m_run, l_run, npages, bh, b, g, s, S, Hkv, lane, ldo, part_o, part_ml, tickets, out);
}

// ------------------------------------------------------------------------------------------------------------------
// Uniform 8-bit KV cache (reference QuantizedKVCache, mlx_vlm/models/cache.py:233-334; its attention
// quantized_scaled_dot_product_attention, models/base.py:260-302; the switch-over maybe_quantize_kv_cache,
// generate/common.py:170-181): mx.quantize(bits = 8, group_size = 64) over the head dimension of every cached key and
// value row - per 64-element group a bf16 scale and bias, w ~ scale * n + bias, n in 0..255.
//
// Pools (one set per layer, pages and block table shared with the bf16 pools):
//   K8  [page][Hkv][D/8][64 keys][8]  u8      the bf16 K layout with 1-byte elements (an S^T A fragment = ONE 8-byte load)
//   V8  [page][Hkv][D][64 key slots]  u8      the bf16 V layout with 1-byte elements
//   KSB / VSB [page][Hkv][64 keys][D/64] u32  (scale bf16 | bias bf16 << 16) of key k, group j at word k * (D/64) + j
// Quantisation (vlm_kv_quantize_tokens; the decode kernel does it for the token the step has just written): the exact
// arithmetic of mx.quantize as oracle/quant.py states it - fp32 min / max of the group, scale = max((max - min) / 255,
// 1e-7) signed so that the edge of larger magnitude lands on an integer, n = clip(rint((w - bias) / scale), 0, 255) from
// the fp32 scale / bias, which are then stored rounded to bf16 - IEEE division and rint throughout: bit-exact with the
// oracle (tests/test_ops_gpu.py).
struct Q8Group { float scale, bias; };
__device__ __forceinline__ Q8Group q8_group_params(float w_max, float w_min) {
  const bool mask = fabsf(w_min) > fabsf(w_max);
  float scale = fmaxf((w_max - w_min) / 255.0f, 1e-7f);
  scale = mask ? scale : -scale;
  const float edge = mask ? w_min : w_max;
  const float q0 = rintf(edge / scale);
  scale = q0 != 0.f ? edge / q0 : scale;
  return Q8Group{scale, q0 == 0.f ? 0.f : edge};
}
__device__ __forceinline__ unsigned q8_value(float w, Q8Group p) {
  return (unsigned)fminf(fmaxf(rintf((w - p.bias) / p.scale), 0.f), 255.f);
}
// one wave quantises the K and V rows of ONE (token, kv head): lane = element of the 64-wide group, both groups of D = 128
__device__ __forceinline__ void q8_quantize_token(const bf16_t* __restrict__ kp16, const bf16_t* __restrict__ vp16,
                                                  unsigned char* kp8, unsigned char* vp8, unsigned* ksb, unsigned* vsb,
                                                  size_t page_head, int within, int lane) {
  // page_head = page * Hkv + g; `within` = token slot in the page
  const bf16_t* kb = kp16 + page_head * (size_t)(HD / 8) * PAGE * 8;
  const bf16_t* vb = vp16 + page_head * (size_t)HD * PAGE;
  unsigned char* kb8 = kp8 + page_head * (size_t)(HD / 8) * PAGE * 8;
  unsigned char* vb8 = vp8 + page_head * (size_t)HD * PAGE;
  const int vs = vlm_vslot(within);
  float kw[2], vw[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int d = 64 * j + lane;
    kw[j] = bf2f(kb[((size_t)(d >> 3) * PAGE + within) * 8 + (d & 7)]);
    vw[j] = bf2f(vb[(size_t)d * PAGE + vs]);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int d = 64 * j + lane;
    const Q8Group pk = q8_group_params(wave_max(kw[j]), -wave_max(-kw[j]));
    const Q8Group pv = q8_group_params(wave_max(vw[j]), -wave_max(-vw[j]));
    kb8[((size_t)(d >> 3) * PAGE + within) * 8 + (d & 7)] = (unsigned char)q8_value(kw[j], pk);
    vb8[(size_t)d * PAGE + vs] = (unsigned char)q8_value(vw[j], pv);
    if (lane == 0) {
      ksb[(page_head * PAGE + within) * 2 + j] = (unsigned)f2bf(pk.scale) | ((unsigned)f2bf(pk.bias) << 16);
      vsb[(page_head * PAGE + within) * 2 + j] = (unsigned)f2bf(pv.scale) | ((unsigned)f2bf(pv.bias) << 16);
    }
  }
}

// KVCache.to_quantized (cache.py:415-423) over the paged pools: token i of the list = slot kv_slot[i] of sequence kv_seq[i]
// (NULL: row i).  One workgroup = 64 consecutive LIST entries of one (kv head, layer), lane = entry: wave 0 / 1 the two
// 64-element groups of K, wave 2 / 3 those of V.  A join quantises whole sequences (slots 0 .. len - 1 in order), so the 64
// lanes of a wave sit on consecutive key slots of a page: K is read as 8 x 16 bytes per lane (1 KiB contiguous per
// instruction across the wave) and V as one 128-byte row per instruction; the group's min / max are IN the lane (64 values),
// no cross-lane step.  (The first version ran one wave per (token, head, layer) with lane = element: 2-byte accesses 128 B
// apart - 1.3 ms per call at 885 x 16 tokens of Phi-3.5, 0.2 TB/s: profiles/r04_phi35v_kv8_kernel_stats.txt.)  Arbitrary
// lists stay legal (every lane resolves its own page); the arithmetic is q8_group_params / q8_value as before: bit-exact.
__global__ __launch_bounds__(256) void kv_quantize_tokens_kernel(const bf16_t* __restrict__ kpool, const bf16_t* __restrict__ vpool,
                                                                 unsigned char* kpool8, unsigned char* vpool8, unsigned* ksb,
                                                                 unsigned* vsb, size_t layer_stride, const int* __restrict__ kv_seq,
                                                                 const int* __restrict__ kv_slot, int T,
                                                                 const int* __restrict__ block_table, int max_pages, int Hkv) {
  const int g = blockIdx.y, layer = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane, ic = min(i, T - 1);
  const bool live = i < T;
  const int seq = kv_seq ? kv_seq[ic] : ic, slot = kv_slot[ic];
  const size_t page = block_table ? (size_t)block_table[(size_t)seq * max_pages + (slot >> 6)] : (size_t)seq * max_pages + (slot >> 6);
  const size_t lo = (size_t)layer * layer_stride;             // elements of one layer's pool (bf16 elements == u8 bytes)
  const size_t page_head = page * Hkv + g;
  const int within = slot & 63, j = wave & 1;
  float w[64];
  if (wave < 2) {
    const bf16_t* kb = kpool + lo + page_head * (size_t)(HD / 8) * PAGE * 8 + ((size_t)(8 * j) * PAGE + within) * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const u32x4_t u = *reinterpret_cast<const u32x4_t*>(kb + (size_t)c * PAGE * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) { w[8 * c + 2 * e] = bf_lo(u[e]); w[8 * c + 2 * e + 1] = bf_hi(u[e]); }
    }
  } else {
    const bf16_t* vb = vpool + lo + page_head * (size_t)HD * PAGE + (size_t)(64 * j) * PAGE + vlm_vslot(within);
#pragma unroll
    for (int d = 0; d < 64; ++d) w[d] = bf2f(vb[(size_t)d * PAGE]);
  }
  float mx = w[0], mn = w[0];
#pragma unroll
  for (int d = 1; d < 64; ++d) { mx = fmaxf(mx, w[d]); mn = fminf(mn, w[d]); }
  const Q8Group pq = q8_group_params(mx, mn);
  if (!live) return;
  if (wave < 2) {
    unsigned char* kb8 = kpool8 + lo + page_head * (size_t)(HD / 8) * PAGE * 8 + ((size_t)(8 * j) * PAGE + within) * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      uint2 o;
      o.x = q8_value(w[8 * c], pq) | (q8_value(w[8 * c + 1], pq) << 8) | (q8_value(w[8 * c + 2], pq) << 16) | (q8_value(w[8 * c + 3], pq) << 24);
      o.y = q8_value(w[8 * c + 4], pq) | (q8_value(w[8 * c + 5], pq) << 8) | (q8_value(w[8 * c + 6], pq) << 16) | (q8_value(w[8 * c + 7], pq) << 24);
      *reinterpret_cast<uint2*>(kb8 + (size_t)c * PAGE * 8) = o;
    }
  } else {
    unsigned char* vb8 = vpool8 + lo + page_head * (size_t)HD * PAGE + (size_t)(64 * j) * PAGE + vlm_vslot(within);
#pragma unroll
    for (int d = 0; d < 64; ++d) vb8[(size_t)d * PAGE] = (unsigned char)q8_value(w[d], pq);
  }
  unsigned* sb = (wave < 2 ? ksb : vsb) + lo / (HD / 2);
  sb[(page_head * PAGE + within) * 2 + j] = (unsigned)f2bf(pq.scale) | ((unsigned)f2bf(pq.bias) << 16);
}

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
// 8 u8 -> 8 fp16 values 1024 + n, ONE v_perm_b32 per two elements: byte n under the constant byte 0x64 is the half-precision
// number 0x64nn = 1024 + n exactly (10 mantissa bits).  The constant 1024 is taken out again in the affine forms below
// (scale * (q . (1024 + n)) + (bias - 1024 scale) * sum(q)), so no subtraction is spent per element - the u8 -> operand
// conversion costs 4 VALU instructions per 8 elements instead of 12 (v_cvt_f32_ubyte + v_cvt_pk_bf16_f32).
__device__ __forceinline__ f16x8_t q8_frag(const u32x2_t w) {
  const unsigned x = w[0], y = w[1], k = 0x64646464u;
  const u32x4_t v = {__builtin_amdgcn_perm(k, x, 0x04010400u), __builtin_amdgcn_perm(k, x, 0x04030402u),
                     __builtin_amdgcn_perm(k, y, 0x04010400u), __builtin_amdgcn_perm(k, y, 0x04030402u)};
  return __builtin_bit_cast(f16x8_t, v);
}
__device__ __forceinline__ unsigned pack_h2(float lo, float hi) {      // two fp32 -> packed fp16x2 (round to nearest even)
  typedef _Float16 h2_t_ __attribute__((ext_vector_type(2)));
  const h2_t_ v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(unsigned, v);
}

// The page-split decode attention over the 8-bit pools.  Arithmetic = the reference's typed graph as far as a split /
// flash formulation allows: q * scale is a typed multiply (rounded to bf16); a score is the fp32 sum over the two groups
// of scale_j * (q . n) + bias_j * sum(q) - the EXACT affine form of quantized_matmul, the integers entering the fp16 MFMA
// (v_mfma_f32_16x16x32_f16) as exact values 1024 + n and q as the exact fp16 image of its bf16 value - rounded to bf16 as
// quantized_matmul's output is; softmax in fp32; P . V with p * scale_v as the fp16 MFMA operand and the bias term
// sum_k p_k * (bias_k - 1024 scale_k) carried in fp32.  (The reference rounds the NORMALISED
// probabilities to bf16; a split kernel rounds the unnormalised ones: same count of roundings, tolerance in the tests.)
// The workgroup whose page holds the step's new token (slot len - 1, written to the bf16 pools by the qkv epilogue)
// quantises it first - QuantizedKVCache.update_and_fetch - and reads it back with the rest of the page (every page load
// is non-temporal: served by L2, behind the workgroup's own drained stores).
// HP ("half pages", the form for many (row, kv head) pairs): a workgroup takes 32 keys of a page instead of 64 - split index
// s' = 2 s + hh covers keys 32 s' .. 32 s' + 31 of every S-th page, so the active splits are exactly s' < ceil(len / 32) and
// the hand-off / merge below runs unchanged over 2 S splits - with half the operand registers (145 instead of 209 VGPRs:
// three waves per SIMD instead of two; the 64-key form of a 512-pair step ran at 3.4 TB/s, latency-bound per wave:
// profiles/r04_phi35v_kv8_kernel_stats.txt).
template <int G, bool IDENT, bool MERGE, bool HP>
__global__ __launch_bounds__(64) void attn_decode_pagesplit_q8_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ kpool16, const bf16_t* __restrict__ vpool16, unsigned char* kpool8,
    unsigned char* vpool8, unsigned* ksb, unsigned* vsb, const int* __restrict__ block_table, const int* __restrict__ kv_len,
    int ldq, int max_pages, int Hkv, int kv_len_add, float scale, int S, int ldo, float* part_o, float* part_ml,
    unsigned* tickets, bf16_t* __restrict__ out, int quantize_new) {
  const int bh = blockIdx.x, b = bh / Hkv, g = bh % Hkv, s2 = blockIdx.y, s = HP ? s2 >> 1 : s2, hh = HP ? s2 & 1 : 0;
  constexpr int NT = HP ? 2 : 4, NU = HP ? 1 : 2;             // key tiles of 16 / 32-key steps per visit
  const int t0 = NT * hh, u0 = hh;
  const int lane = threadIdx.x, head = lane & 15, gq = lane >> 4;
  const int len = kv_len[b] + kv_len_add, npages = (len + PAGE - 1) / PAGE;
  const int* trow = IDENT ? nullptr : block_table + (size_t)b * max_pages;
  // Q fragments with the reference's typed q * scale, and the per-group sums of q the bias terms need
  f16x8_t qf[4];
  float sq[2] = {0.f, 0.f};
  {
    const bf16_t* qr = q + (size_t)b * ldq + (size_t)(g * G + min(head, G - 1)) * HD + 8 * gq;
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
      const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qr + 32 * ds);
      u32x4_t sc;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned w = raw[j];
        const float a = rbf(bf_lo(w) * scale), c = rbf(bf_hi(w) * scale);      // the typed q * scale (bf16)
        sc[j] = pack_h2(a, c);                                                 // exact in fp16 (8 significant bits)
        sq[ds >> 1] += a + c;
      }
      qf[ds] = __builtin_bit_cast(f16x8_t, sc);
    }
    sq[0] = col4_sum(sq[0]);            // over the four 8-wide d chunks a head's lanes hold per 32-step
    sq[1] = col4_sum(sq[1]);
  }
  f32x4_t ot[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ot[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f, ob[2] = {0.f, 0.f};
  constexpr float LOG2E = 1.44269504088896340736f;
  for (int pc = s; pc < npages; pc += S) {
    if (HP && pc * PAGE + 32 * hh >= len) break;          // the last page's second half may be empty
    const size_t page = IDENT ? (size_t)b * max_pages + pc : (size_t)trow[pc];
    const size_t ph = page * Hkv + g;
    if (quantize_new && pc == (len - 1) / PAGE && (!HP || hh == (((len - 1) & 63) >> 5))) {
      q8_quantize_token(kpool16, vpool16, kpool8, vpool8, ksb, vsb, ph, (len - 1) & 63, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned char* kp = kpool8 + ph * (size_t)(HD / 8) * PAGE * 8 + ((size_t)gq * PAGE + head) * 8;
    const unsigned char* vp = vpool8 + ph * (size_t)HD * PAGE + (size_t)head * PAGE + 8 * gq;
    const unsigned* ks = ksb + (ph * PAGE + 4 * gq) * 2;
    const unsigned* vs = vsb + (ph * PAGE + 4 * gq) * 2;
    u32x2_t kf[NT][4], vf[8][NU];
    u32x4_t kq[NT][2], vq[NT][2];         // (scale | bias) words of keys 16 t + 4 gq + r: [t][half]: r = 2 half, 2 half + 1 x 2 groups
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int ds = 0; ds < 4; ++ds)
        kf[tt][ds] = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(kp + ((size_t)(4 * ds) * PAGE + 16 * (t0 + tt)) * 8));
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        kq[tt][hf] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(ks + (16 * (t0 + tt)) * 2 + 4 * hf));
        vq[tt][hf] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vs + (16 * (t0 + tt)) * 2 + 4 * hf));
      }
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
      for (int uu = 0; uu < NU; ++uu)
        vf[dt][uu] = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(vp + (size_t)(16 * dt) * PAGE + 32 * (u0 + uu)));
    __builtin_amdgcn_sched_barrier(0);
    // ---- S^T: per key tile two group accumulators, then the affine form per (key, group)
    float sc[NT][4];
    float mt = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q8_frag(kf[t][0]), qf[0], a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q8_frag(kf[t][1]), qf[1], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q8_frag(kf[t][2]), qf[2], a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q8_frag(kf[t][3]), qf[3], a1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const u32x4_t w4 = kq[t][r >> 1];
        const unsigned w0 = (r & 1) ? w4[2] : w4[0], w1 = (r & 1) ? w4[3] : w4[1];     // groups 0 / 1 of key 16 t + 4 gq + r
        const int key = pc * PAGE + 16 * (t0 + t) + 4 * gq + r;
        // quantized_matmul: fp32 sum over the dequantised keys, one rounding to the query dtype
        // (a_j = q . (1024 + n): the constant leaves through the bias factor)
        const float sv = rbf(bf_lo(w0) * a0[r] + (bf_hi(w0) - 1024.f * bf_lo(w0)) * sq[0] +
                             bf_lo(w1) * a1[r] + (bf_hi(w1) - 1024.f * bf_lo(w1)) * sq[1]);
        sc[t][r] = key < len ? sv * LOG2E : -INFINITY;
        mt = fmaxf(mt, sc[t][r]);
      }
    }
    mt = col4_max(mt);
    const float m_new = fmaxf(m_run, mt);
    const float alpha = exp2f(m_run - m_new);      // (every page processed holds a valid key: m_new is finite)
    float ls = 0.f, pbias[2] = {0.f, 0.f};
    u32x4_t pk[2][NU];                             // P'^T fragments [group][u]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float pp[2][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = pc * PAGE + 16 * (t0 + t) + 4 * gq + r;
        const bool ok = key < len;
        const float pr = ok ? rbf(exp2f(sc[t][r] - m_new)) : 0.f;     // the probabilities are a bf16 tensor in the reference
        ls += pr;
        const u32x4_t w4 = vq[t][r >> 1];
        const unsigned w0 = (r & 1) ? w4[2] : w4[0], w1 = (r & 1) ? w4[3] : w4[1];
        pp[0][r] = ok ? pr * bf_lo(w0) : 0.f;
        pp[1][r] = ok ? pr * bf_lo(w1) : 0.f;
        pbias[0] += ok ? pr * (bf_hi(w0) - 1024.f * bf_lo(w0)) : 0.f;         // the V operands are 1024 + n as well
        pbias[1] += ok ? pr * (bf_hi(w1) - 1024.f * bf_lo(w1)) : 0.f;
      }
      // k-slot 8 gq + j of step u <- tile 2u (j < 4) / tile 2u + 1 (j >= 4): tile t fills words (t & 1) * 2, + 1 of step t >> 1
#pragma unroll
      for (int gI = 0; gI < 2; ++gI) {
        pk[gI][t >> 1][(t & 1) * 2] = pack_h2(pp[gI][0], pp[gI][1]);
        pk[gI][t >> 1][(t & 1) * 2 + 1] = pack_h2(pp[gI][2], pp[gI][3]);
      }
    }
    l_run = l_run * alpha + ls;
    ob[0] = ob[0] * alpha + pbias[0];
    ob[1] = ob[1] * alpha + pbias[1];
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;
#pragma unroll
      for (int u = 0; u < NU; ++u)
        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q8_frag(vf[dt][u]), __builtin_bit_cast(f16x8_t, pk[dt >> 2][u]), ot[dt], 0, 0, 0);
    }
  }
  // the bias terms: one scalar per (head, group), the same for every d of the group
  ob[0] = col4_sum(ob[0]);
  ob[1] = col4_sum(ob[1]);
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) ot[dt][r] += ob[dt >> 2];
  // (HP: 2 S splits of half pages; the active ones are s2 < ceil(len / 32))
  pagesplit_finish<G, MERGE>(ot, m_run, l_run, HP ? (len + 31) / 32 : npages, bh, b, g, s2, HP ? 2 * S : S, Hkv, lane, ldo, part_o,
                             part_ml, tickets, out);
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

extern "C" int vlm_attn_decode_paged(const void* q, int ldq, const void* kpool, const void* vpool, const void* block_table,
                                     int max_pages, const void* kv_len, int kv_len_add, int B, int Hq, int Hkv, int D, float scale,
                                     int nsplit, void* part_o, void* part_ml, void* out, int ldo, void* stream) {
  if (!q || !kpool || !vpool || !kv_len || max_pages <= 0) return VLM_ERR_ARG;   // block_table == NULL: identity layout
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || nsplit <= 0 || Hq % Hkv != 0) return VLM_ERR_ARG;
  if (nsplit > 1 && (!part_o || !part_ml)) return VLM_ERR_ARG;
  if (nsplit == 1 && !out) return VLM_ERR_ARG;
  if (D != HD || ldq % 8 != 0) return VLM_ERR_SHAPE;
  const int G = Hq / Hkv;
  hipStream_t st = (hipStream_t)stream;
  const float sl2 = scale * 1.44269504088896340736f;
  dim3 grid(B * Hkv, nsplit);
  bf16_t* direct = nsplit == 1 ? (bf16_t*)out : nullptr;
#ifdef VLM_ATTN_TIMELINE   // debug build only (-DVLM_ATTN_TIMELINE): wave timeline stamps overwrite part_o
  static const bool stamps_env = getenv("VLM_ATTN_STAMPS") != nullptr;
  const bool stamps = stamps_env && part_o != nullptr && block_table != nullptr;
#else
  const bool stamps = false;
#endif   // debug timeline
#define GO(GV)                                                                                                          \
  if (stamps)                                                                                                           \
    hipLaunchKernelGGL((attn_decode_mfma_kernel<GV, 8, true, false>), grid, dim3(8 * 64), 0, st, \
                       (const bf16_t*)q, (const bf16_t*)kpool, (const bf16_t*)vpool, (const int*)block_table,           \
                       (const int*)kv_len, ldq, max_pages, Hkv, kv_len_add, Hq, sl2, nsplit, ldo, (float*)part_o,       \
                       (float*)part_ml, direct);                                                                  \
  else                                                                                                                  \
  if (!block_table)                                                                                                     \
    hipLaunchKernelGGL((attn_decode_mfma_kernel<GV, 8, false, true>), grid, dim3(8 * 64), 0, st, (const bf16_t*)q,      \
                       (const bf16_t*)kpool, (const bf16_t*)vpool, (const int*)nullptr, (const int*)kv_len, ldq,        \
                       max_pages, Hkv, kv_len_add, Hq, sl2, nsplit, ldo, (float*)part_o, (float*)part_ml, direct);\
  else                                                                                                                  \
    hipLaunchKernelGGL((attn_decode_mfma_kernel<GV, 8, false, false>), grid, dim3(8 * 64), 0, st, (const bf16_t*)q,     \
                       (const bf16_t*)kpool, (const bf16_t*)vpool, (const int*)block_table, (const int*)kv_len, ldq,    \
                       max_pages, Hkv, kv_len_add, Hq, sl2, nsplit, ldo, (float*)part_o, (float*)part_ml, direct)
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
  if (nsplit > 1 && out) {   // otherwise the caller merges the splits itself (vlm_gemv_attn_out)
    hipLaunchKernelGGL(attn_decode_combine_kernel, dim3(B * Hq), dim3(HD), 0, st, (const float*)part_o,
                       (const float*)part_ml, nsplit, (bf16_t*)out, ldo, Hq);
    VLM_CHECK_LAUNCH();
  }
  return VLM_OK;
}

extern "C" int vlm_attn_decode_paged_split(const void* q, int ldq, const void* kpool, const void* vpool,
                                           const void* block_table, int max_pages, const void* kv_len, int kv_len_add, int B,
                                           int Hq, int Hkv, int D, float scale, int nsplit, void* part_o, void* part_ml,
                                           void* tickets, void* out, int ldo, void* stream) {
  if (!q || !kpool || !vpool || !kv_len || !part_o || !part_ml || max_pages <= 0) return VLM_ERR_ARG;
  if (out && !tickets) return VLM_ERR_ARG;
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || nsplit <= 0 || nsplit > 65535 || Hq % Hkv != 0) return VLM_ERR_ARG;
  if (D != HD || ldq % 8 != 0 || ldo % 4 != 0) return VLM_ERR_SHAPE;
  if ((size_t)B * Hq * nsplit * HD * 4 >= ((size_t)1 << 31)) return VLM_ERR_SHAPE;   // 32-bit buffer offsets
  const int G = Hq / Hkv;
  hipStream_t st = (hipStream_t)stream;
  const float sl2 = scale * 1.44269504088896340736f;
  dim3 grid(B * Hkv, nsplit);
#define GO1(GV, ID, MG)                                                                                                  \
  hipLaunchKernelGGL((attn_decode_pagesplit_kernel<GV, ID, MG>), grid, dim3(64), 0, st, (const bf16_t*)q,                 \
                     (const bf16_t*)kpool, (const bf16_t*)vpool, (const int*)block_table, (const int*)kv_len, ldq, max_pages, \
                     Hkv, kv_len_add, sl2, nsplit, ldo, (float*)part_o, (float*)part_ml, (unsigned*)tickets, (bf16_t*)out, 0)
#define GO(GV)                                                                                                           \
  do {                                                                                                                   \
    if (!block_table) { if (out) GO1(GV, true, true); else GO1(GV, true, false); }                                       \
    else { if (out) GO1(GV, false, true); else GO1(GV, false, false); }                                                  \
  } while (0)
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
#undef GO1
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_kv_quantize_tokens(const void* kpool, const void* vpool, void* kpool8, void* vpool8, void* ksb, void* vsb,
                                      size_t layer_stride, int n_layers, const void* kv_seq, const void* kv_slot, int T,
                                      const void* block_table, int max_pages, int Hkv, int D, void* stream) {
  if (!kpool || !vpool || !kpool8 || !vpool8 || !ksb || !vsb || !kv_slot || n_layers <= 0 || Hkv <= 0 || max_pages <= 0 || T < 0)
    return VLM_ERR_ARG;
  if (D != HD) return VLM_ERR_SHAPE;
  if (T == 0) return VLM_OK;
  if (T > 65535 * 1024 || Hkv > 65535 || n_layers > 65535) return VLM_ERR_SHAPE;
  hipLaunchKernelGGL(kv_quantize_tokens_kernel, dim3((T + 63) / 64, Hkv, n_layers), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)kpool, (const bf16_t*)vpool, (unsigned char*)kpool8, (unsigned char*)vpool8, (unsigned*)ksb,
                     (unsigned*)vsb, layer_stride, (const int*)kv_seq, (const int*)kv_slot, T, (const int*)block_table, max_pages,
                     Hkv);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_attn_decode_paged_q8(const void* q, int ldq, const void* kpool16, const void* vpool16, void* kpool8,
                                        void* vpool8, void* ksb, void* vsb, const void* block_table, int max_pages,
                                        const void* kv_len, int kv_len_add, int B, int Hq, int Hkv, int D, float scale,
                                        int nsplit, void* part_o, void* part_ml, void* tickets, void* out, int ldo,
                                        int quantize_new, void* stream) {
  if (!q || !kpool8 || !vpool8 || !ksb || !vsb || !kv_len || !part_o || !part_ml || max_pages <= 0) return VLM_ERR_ARG;
  if (quantize_new && (!kpool16 || !vpool16)) return VLM_ERR_ARG;
  if (out && !tickets) return VLM_ERR_ARG;
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || nsplit <= 0 || nsplit > 65535 || Hq % Hkv != 0) return VLM_ERR_ARG;
  if (D != HD || ldq % 8 != 0 || ldo % 4 != 0) return VLM_ERR_SHAPE;
  if ((size_t)B * Hq * nsplit * HD * 4 >= ((size_t)1 << 31)) return VLM_ERR_SHAPE;
  const int G = Hq / Hkv;
  hipStream_t st = (hipStream_t)stream;
  // `queries *= scale` (base.py:272) with a python float: MLX converts the weak scalar to the ARRAY's dtype first, so the
  // typed multiply uses bf16(scale) (128 ** -0.5 -> 0.08837890625) - pinned by tests/golden/kvquant_ref.npz, where the
  // reference's own function runs; round-to-nearest-even on the host
  {
    unsigned u;
    memcpy(&u, &scale, 4);
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    memcpy(&scale, &u, 4);
  }
  // half-page units once the (row, kv head) pairs alone give every SIMD its waves (Phi-3.5 at 16 rows: 512 pairs): the 64-key
  // form holds 209 registers (2 waves per SIMD); VLM_ATTN_Q8_HALF = 0 / 1 forces (A/B)
  static const int hp_env = [] { const char* e = getenv("VLM_ATTN_Q8_HALF"); return e ? atoi(e) : -1; }();
  const bool hp = (hp_env >= 0 ? hp_env != 0 : B * Hkv >= 128) && nsplit <= 16 && out != nullptr;   // (merging form only)
  dim3 grid_q8(B * Hkv, hp ? 2 * nsplit : nsplit);
#define GO2(GV, ID, MG, HP)                                                                                              \
  hipLaunchKernelGGL((attn_decode_pagesplit_q8_kernel<GV, ID, MG, HP>), grid_q8, dim3(64), 0, st, (const bf16_t*)q,       \
                     (const bf16_t*)kpool16, (const bf16_t*)vpool16, (unsigned char*)kpool8, (unsigned char*)vpool8,      \
                     (unsigned*)ksb, (unsigned*)vsb, (const int*)block_table, (const int*)kv_len, ldq, max_pages, Hkv,    \
                     kv_len_add, scale, nsplit, ldo, (float*)part_o, (float*)part_ml, (unsigned*)tickets, (bf16_t*)out,   \
                     quantize_new)
#define GO1(GV, ID, MG)                  \
  do {                                   \
    if (hp) GO2(GV, ID, MG, true);       \
    else GO2(GV, ID, MG, false);         \
  } while (0)
#define GO(GV)                                                                                                           \
  do {                                                                                                                   \
    if (!block_table) { if (out) GO1(GV, true, true); else GO1(GV, true, false); }                                       \
    else { if (out) GO1(GV, false, true); else GO1(GV, false, false); }                                                  \
  } while (0)
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
#undef GO1
#undef GO2
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
