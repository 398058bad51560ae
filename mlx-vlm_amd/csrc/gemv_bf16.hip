// Decode-time weight-streaming GEMV family for gfx950 (HBM-bound):
//     y[m][n] = epilogue( sum_k prologue(x)[m][k] * W[n][k] ),   m < MB <= 8
//
// Replaces the nn.Linear calls of the reference's per-token decode step
// (mlx_vlm/models/qwen2_vl/language.py:52-55,76,120 q/k/v/o projections;
//  mlp.py:9-14 + activations.py:7-9 SwiGLU MLP; language.py:514-517 lm_head /
//  embed_tokens.as_linear) with their neighbours fused in:
//    prologues  RMSNorm of the residual stream (language.py:130-133,149-153,200);
//               merge of the split-K attention partials (the tail of
//               mx.fast.scaled_dot_product_attention, base.py:366-373)
//    epilogues  bias; residual add (language.py:151-153); SwiGLU on interleaved
//               gate/up rows; M-RoPE (fused-kernel numerics, rope_utils.py:567-651)
//               + paged KV write (KVCache.update_and_fetch, cache.py:345-367)
// so a decoder layer is 5 launches: [norm+qkv+rope+kv] [attention] [merge+o_proj+res]
// [norm+gate/up+swiglu] [down+res].
//
// Design (CDNA4).  No LDS round trip for weights: each lane streams 16-byte
// non-temporal loads of its weight rows straight into VGPRs, and ALL of a wave's
// weight loads are issued BEFORE the prologue runs, so the HBM stream overlaps
// the (latency-bound) norm / merge prologue instead of waiting behind it.
//   * row-wave kernel (K <= 3584): one wave owns R full rows; the activation
//     vector is produced once per workgroup into LDS; v_dot2c_f32_bf16 in fp32;
//     wavefront xor-shuffle reductions.
//   * split-K kernel (large K, e.g. the 8960-wide down projection): the 4 waves of
//     a workgroup split K for RW rows (N / RW workgroups keep all 256 CUs busy),
//     partial sums meet in LDS.
#include "common.hpp"
#include "attn_pagesplit.hpp"
#include "internal.h"
#include "../../include/vlm_hip.h"

namespace {

enum { PRO_NONE = 0, PRO_RMSNORM = 1, PRO_ATTN = 2, PRO_ATTN_PS = 3 };
constexpr int EPI_ROPE_KV = 1 << 10;   // internal epilogue id (qkv projection)

struct RopeKvArgs {
  const int* pos;            // [M] rope position (all three M-RoPE axes are equal for a decoded text token)
  const int* slot;           // [M] KV slot (= tokens already in the cache)
  const float* inv_freq;     // [D/2]
  const int* block_table;    // [M][max_pages]
  int max_pages, Hq, Hkv, D;
  bf16_t* kpool;             // [page][Hkv][D/8][64][8]
  bf16_t* vpool;             // [page][Hkv][D][64 key slots]
  float qk_scale = 1.f;      // q and k are multiplied by this (a typed op: rounded to bf16) before the rotation - SuScaledRoPE
  int long_from = 0;         // > 0: inv_freq holds [2][D/2] (short, long factors); the LONG row applies to every row of the step
                             // when ANY row's slot (= cache offset) >= long_from: SuScaledRoPE's per-call rule
                             // (rope_utils.py:168-172: position_end = max(offset) + tokens of the call > original_max)
};

struct AttnProArgs {
  const float* part_o;       // [M][Hq][S][D]  fp32 (PRO_ATTN_PS: the page-split attention's partials)
  const float* part_ml;      // [M][Hq][S][2]  (PRO_ATTN: m in the natural-log domain; PRO_ATTN_PS: log2 domain, -inf = no page)
  int S, Hq, D;
};
constexpr int AT2_S = VLM_MERGE_S;    // splits the PRO_ATTN_PS prologue reads (all of them, unconditionally)

__device__ __forceinline__ u32x4_t ntl(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); }

// NOTE: __builtin_bit_cast on a vector-ELEMENT lvalue (w.y) silently reads element 0 with this
// compiler (hipcc / clang 22, verified in the .s: four identical v_dot2c); copy to a scalar first.
__device__ __forceinline__ float dot2(unsigned w, unsigned x, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, x), acc, false);
}
__device__ __forceinline__ float dot8(const u32x4_t w, const u32x4_t x, float acc) {
  const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
  acc = dot2(w0, x0, acc);
  acc = dot2(w1, x1, acc);
  acc = dot2(w2, x2, acc);
  acc = dot2(w3, x3, acc);
  return acc;
}

// ------------------------------------------------------------------------------------------
// row-wave kernel: K <= 512 * KC
// ------------------------------------------------------------------------------------------
template <int R, int KC, int MB, int PRO, int EPI>
__global__ __launch_bounds__(256) void gemv_rowwave_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                           const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                           const bf16_t* __restrict__ norm_w, bf16_t* __restrict__ y, int N,
                                                           int K, int ldx, int ldw, int ldy, int ldres, float eps,
                                                           RopeKvArgs rk, AttnProArgs ap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // xs[MB][K] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = K >> 3;
  const int gw = blockIdx.x * 4 + wave;

  // ---- which rows does this wave own?
  int row[R];
  bool rope_pair = false;
  int rope_head = 0, rope_j = 0;
  if (EPI == EPI_ROPE_KV) {
    const int half = rk.D >> 1, n_pair = (rk.Hq + rk.Hkv) * half;
    if (gw < n_pair) {
      rope_pair = true;
      rope_head = gw / half;
      rope_j = gw % half;
      row[0] = rope_head * rk.D + rope_j;
      row[R - 1] = row[0] + half;
    } else {
      row[0] = (rk.Hq + rk.Hkv) * rk.D + 2 * (gw - n_pair);
      row[R - 1] = row[0] + 1;
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) row[r] = gw * R + r;
  }
  const bool active = row[0] < N;

  // ---- loads are returned in issue order (vmcnt), so the SMALL latency-critical ones go first: the
  //      activation row for the norm prologue and the epilogue operands; then every weight load of the
  //      wave.  The prologue then only waits for its own (oldest) loads while the weights keep streaming.
  u32x4_t xv[KC], xv2[MB > 4 ? KC : 1];
  uint4 nwv[KC];
  if (PRO == PRO_RMSNORM) {
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      // branch-free (clamped address + select on the data): a load under a divergent branch makes hipcc
      // fall back to s_waitcnt vmcnt(0) instead of a counted wait
      const int ch = lane + 64 * c, chc = min(ch, nchunk - 1);
      xv[c] = *reinterpret_cast<const u32x4_t*>(x + (size_t)(wave % MB) * ldx + (size_t)chc * 8);   // masked at use
      nwv[c] = reinterpret_cast<const uint4*>(norm_w)[chc];
      // MB == 8: this wave also normalises row 4 + wave; its loads go out HERE, ahead of the weight stream (issued
      // after it they would return behind every weight load of the wave - vector loads return in order)
      if (MB > 4) xv2[c] = *reinterpret_cast<const u32x4_t*>(x + (size_t)(4 + wave) * ldx + (size_t)chc * 8);
    }
  }
  // PRO_ATTN fast path (<= 4 splits, MB <= 2): the split merge is one 8-element chunk of x per thread; its loads
  // (m/l pairs + the partial O rows) are independent of everything, so they go out here, ahead of the weight stream
  // - the merge then costs no extra memory round trip (the first version merged through LDS in three dependent
  // phases and made this kernel 8 us slower than the plain o_proj)
  constexpr int AT_IT = (PRO == PRO_ATTN && MB <= 2) ? MB : 1;     // chunks per thread: MB * K / 8 <= 256 * MB
  const bool at_fast = PRO == PRO_ATTN && MB <= 2 && ap.S <= 4;
  float4 at_o[AT_IT][4][2];
  float2 at_ml[AT_IT][4];
  if (PRO == PRO_ATTN && MB <= 2) {
    if (at_fast) {
#pragma unroll
      for (int it = 0; it < AT_IT; ++it) {
        const int item = min(tid + 256 * it, MB * nchunk - 1), m = item / nchunk, hd = (item % nchunk) * 8;
        const size_t base = ((size_t)m * ap.Hq + hd / ap.D) * ap.S;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
          const size_t e = base + min(sp, ap.S - 1);
          at_ml[it][sp] = *reinterpret_cast<const float2*>(ap.part_ml + e * 2);
          const float4* po = reinterpret_cast<const float4*>(ap.part_o + e * ap.D + hd % ap.D);
          at_o[it][sp][0] = po[0];
          at_o[it][sp][1] = po[1];
        }
      }
    }
  }
  // PRO_ATTN_PS (M == 1, K <= 2048, S <= 16): the merge of the page-split decode attention (attn_decode.hip, MERGE =
  // false) - one 8-element chunk of x per thread, EVERY split's (m, l) and fp32 O chunk loaded unconditionally (a split
  // without a page carries m = -inf and is dropped by a select; its O bytes may be anything), all of it issued here, ahead
  // of the weight stream: the attention launch ends at its partial stores (no ticket, no last-arriver pass), and this
  // kernel pays S * 40 B per thread of L2 reads in front of its weights
  float2 a2_ml[PRO == PRO_ATTN_PS ? AT2_S : 1];
  f32x4_t a2_o[PRO == PRO_ATTN_PS ? AT2_S : 1][2];
  if (PRO == PRO_ATTN_PS) {
    const int ch = min(tid, nchunk - 1), hd0 = ch * 8;
    const size_t base = (size_t)(hd0 / ap.D) * ap.S;
    const float* po = ap.part_o;
#pragma unroll
    for (int sp = 0; sp < AT2_S; ++sp) {
      const size_t e = base + min(sp, ap.S - 1);
      a2_ml[sp] = *reinterpret_cast<const float2*>(ap.part_ml + e * 2);
      a2_o[sp][0] = *reinterpret_cast<const f32x4_t*>(po + e * ap.D + hd0 % ap.D);
      a2_o[sp][1] = *reinterpret_cast<const f32x4_t*>(po + e * ap.D + hd0 % ap.D + 4);
    }
  }
  // epilogue operands (bias / residual / rope position, slot, page) for the lane that will store
  // (raw values only - any arithmetic on them here would force a wait before the weight loads are issued)
  bf16_t e_b0 = 0, e_b1 = 0, e_r = 0;
  int e_pos = 0, e_slot = 0;
  float e_if = 0.f, e_if2 = 0.f;
  if (EPI == EPI_ROPE_KV) {
    const int mm = min(lane, MB - 1), r0 = min(row[0], N - 1), r1 = min(row[R - 1], N - 1);
    e_b0 = bias[r0];
    e_b1 = bias[r1];
    e_slot = rk.slot[mm];
    e_pos = rk.pos[mm];
    e_if = rk.inv_freq[rope_j];
    e_if2 = rk.inv_freq[(rk.long_from > 0 ? (rk.D >> 1) : 0) + rope_j];       // (the same word again when there is one table)
  } else if (!(EPI & VLM_EPI_SWIGLU)) {
    const int ll = min(lane, R * MB - 1), r = ll / MB, m = ll % MB;
    int rr = row[0];
#pragma unroll
    for (int q = 1; q < R; ++q) rr = (r == q) ? row[q] : rr;
    rr = min(rr, N - 1);
    if (EPI & VLM_EPI_BIAS) e_b0 = bias[rr];
    if (EPI & VLM_EPI_RESIDUAL) e_r = res[(size_t)m * ldres + rr];
  }

  __builtin_amdgcn_sched_barrier(0);   // keep the issue order: small loads, THEN the weight stream, THEN math
  u32x4_t wv[R][KC];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bf16_t* wr = W + (size_t)min(row[r], N - 1) * ldw;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const int ch = lane + 64 * c;
      wv[r][c] = ntl(wr + (size_t)min(ch, nchunk - 1) * 8);   // clamped; the x chunk is masked instead
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: activation vector(s) -> LDS as bf16
  if (PRO == PRO_RMSNORM) {
    // every wave normalises row (wave % MB) - straight-line code, so the x loads above stay ahead of the
    // weight loads and get a counted wait; waves >= MB duplicate the (tiny) work and just do not store
#pragma unroll
    for (int round = 0; round < (MB + 3) / 4; ++round) {
      const int m = round * 4 + (round == 0 ? wave % MB : wave);
      if (round > 0) {   // MB == 8: rows 4..7 (loaded up front)
#pragma unroll
        for (int c = 0; c < KC; ++c) xv[c] = xv2[MB > 4 ? c : 0];
      }
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        if (lane + 64 * c >= nchunk) xv[c] = u32x4_t{0, 0, 0, 0};
        const float v[8] = {bf_lo(xv[c][0]), bf_hi(xv[c][0]), bf_lo(xv[c][1]), bf_hi(xv[c][1]),
                            bf_lo(xv[c][2]), bf_hi(xv[c][2]), bf_lo(xv[c][3]), bf_hi(xv[c][3])};
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j] * v[j];
      }
      const float inv = rsqrtf(wave_sum(s) / (float)K + eps);
      uint4* xs = reinterpret_cast<uint4*>(smem + (size_t)m * K * 2);
      const bool writer = round > 0 || wave < MB;
      // pin the norm-weight registers here: otherwise hipcc sinks their loads into the store branch below,
      // behind the weight stream, and waits vmcnt(0) for them
#pragma unroll
      for (int c = 0; c < KC; ++c) asm volatile("" : "+v"(nwv[c].x), "+v"(nwv[c].y), "+v"(nwv[c].z), "+v"(nwv[c].w));
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int ch = lane + 64 * c;
        const u32x4_t u = xv[c];
        const uint4 wu = nwv[c];   // used unconditionally so its load is not sunk into the branch below
        uint4 o;
        o.x = pack_bf2(bf_lo(wu.x) * rbf(bf_lo(u[0]) * inv), bf_hi(wu.x) * rbf(bf_hi(u[0]) * inv));
        o.y = pack_bf2(bf_lo(wu.y) * rbf(bf_lo(u[1]) * inv), bf_hi(wu.y) * rbf(bf_hi(u[1]) * inv));
        o.z = pack_bf2(bf_lo(wu.z) * rbf(bf_lo(u[2]) * inv), bf_hi(wu.z) * rbf(bf_hi(u[2]) * inv));
        o.w = pack_bf2(bf_lo(wu.w) * rbf(bf_lo(u[3]) * inv), bf_hi(wu.w) * rbf(bf_hi(u[3]) * inv));
        if (writer && ch < nchunk) xs[ch] = o;
      }
    }
  } else if (PRO == PRO_ATTN && MB <= 2 && at_fast) {
    // x[m][h*D + d] = sum_s f_s O_s[d],  f_s = e^{m_s - M} / sum_t e^{m_t - M} l_t   (merge of the attention splits)
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
    for (int it = 0; it < AT_IT; ++it) {
      const int item = tid + 256 * it;
      float mm = -INFINITY;
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) mm = fmaxf(mm, sp < ap.S ? at_ml[it][sp].x : -INFINITY);
      float f[4], ll = 0.f;
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        f[sp] = (sp < ap.S && at_ml[it][sp].x != -INFINITY) ? __expf(at_ml[it][sp].x - mm) : 0.f;
        ll += f[sp] * at_ml[it][sp].y;
      }
      const float il = 1.0f / ll;
      float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const float fs = f[sp] * il;
        const float4 a = at_o[it][sp][0], b = at_o[it][sp][1];
        acc8[0] += fs * a.x; acc8[1] += fs * a.y; acc8[2] += fs * a.z; acc8[3] += fs * a.w;
        acc8[4] += fs * b.x; acc8[5] += fs * b.y; acc8[6] += fs * b.z; acc8[7] += fs * b.w;
      }
      if (item < MB * nchunk) {
        uint4 o;
        o.x = pack_bf2(acc8[0], acc8[1]); o.y = pack_bf2(acc8[2], acc8[3]);
        o.z = pack_bf2(acc8[4], acc8[5]); o.w = pack_bf2(acc8[6], acc8[7]);
        reinterpret_cast<uint4*>(xs)[item] = o;     // item = m * nchunk + chunk: xs is [MB][K]
      }
    }
  } else if (PRO == PRO_ATTN_PS) {
    // x[h*D + d] = sum_s f_s O_s[d] / sum_s f_s l_s,  f_s = 2^(m_s - M)   (attn_pagesplit.hpp)
    const uint4 o = vlm_merge_splits16(a2_ml, a2_o, ap.S);
    if (tid < nchunk) reinterpret_cast<uint4*>(smem)[tid] = o;
  } else if (PRO == PRO_ATTN) {
    // general form (more splits / rows): three short phases so that every global load of a phase is independent
    // (one memory round trip each)
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    float* mls = reinterpret_cast<float*>(smem + (size_t)MB * K * 2);   // [MB*Hq][S][2] -> f in slot 0
    const int HD = ap.Hq * ap.D, nml = MB * ap.Hq * ap.S * 2;
    for (int i = tid; i < nml; i += 256) mls[i] = ap.part_ml[i];
    __syncthreads();
    if (tid < MB * ap.Hq) {
      float* r = mls + (size_t)tid * ap.S * 2;
      float mm = -INFINITY;
      for (int s = 0; s < ap.S; ++s) mm = fmaxf(mm, r[2 * s]);
      float ll = 0.f;
      for (int s = 0; s < ap.S; ++s) ll += (r[2 * s] == -INFINITY) ? 0.f : __expf(r[2 * s] - mm) * r[2 * s + 1];
      const float il = 1.0f / ll;
      for (int s = 0; s < ap.S; ++s) r[2 * s] = (r[2 * s] == -INFINITY) ? 0.f : __expf(r[2 * s] - mm) * il;
    }
    __syncthreads();
    for (int i = tid; i < MB * HD; i += 256) {
      const int m = i / HD, hd = i % HD, h = hd / ap.D, d = hd % ap.D;
      const size_t base = ((size_t)m * ap.Hq + h) * ap.S;
      const float* f = mls + base * 2;
      float acc = 0.f;
#pragma unroll 8
      for (int s = 0; s < ap.S; ++s) acc += f[2 * s] * ap.part_o[(base + s) * ap.D + d];
      xs[(size_t)m * K + hd] = f2bf(acc);
    }
  } else {
    for (int i = tid; i < MB * nchunk; i += 256) {
      const int m = i / nchunk, ch = i % nchunk;
      reinterpret_cast<uint4*>(smem + (size_t)m * K * 2)[ch] = reinterpret_cast<const uint4*>(x + (size_t)m * ldx)[ch];
    }
  }
  // the rotation's sine / cosine depend on the position and the frequency table only: computed HERE, while the weight
  // stream is still in flight (the precise sincosf is a few hundred instructions - at the tail of the kernel it sat on
  // the critical path of every qkv launch)
  float rope_sn = 0.f, rope_cs = 1.f;
  bool use_long = false;
  if (EPI == EPI_ROPE_KV) {
    // lanes >= MB hold row MB - 1's slot again: the vote covers exactly the rows of the step (wave-uniform result)
    use_long = rk.long_from > 0 && __any(e_slot >= rk.long_from);
    sincosf((float)e_pos * (use_long ? e_if2 : e_if), &rope_sn, &rope_cs);
  }
  __syncthreads();
  if (!active) return;

  float acc[R][MB];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunk) {
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(smem + ((size_t)m * K + (size_t)ch * 8) * 2);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][m] = dot8(wv[r][c], xv, acc[r][m]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[r][m] = wave_sum(acc[r][m]);

  if (EPI == EPI_ROPE_KV) {
    // R == 2: (d, d + D/2) of one q/k head, or two consecutive v rows.  Lane m stores batch row m.
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      a0 = (lane == m) ? acc[0][m] : a0;
      a1 = (lane == m) ? acc[R - 1][m] : a1;
    }
    if (lane < MB) {
      const int m = lane;
      const float y0 = rbf(a0 + bf2f(e_b0)), y1 = rbf(a1 + bf2f(e_b1));
      // block_table == NULL: identity layout (sequence m owns pages [m * max_pages, (m + 1) * max_pages)) - no
      // dependent table load at the tail of the kernel
      const size_t e_page = rk.block_table ? (size_t)rk.block_table[(size_t)m * rk.max_pages + (e_slot >> 6)]
                                           : (size_t)m * rk.max_pages + (e_slot >> 6);
      const int e_within = e_slot & 63;
      if (rope_pair) {
        const float sn = rope_sn, cs = rope_cs;
        const float z0 = rbf(y0 * rk.qk_scale), z1 = rbf(y1 * rk.qk_scale);      // (exact no-op at scale 1)
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
        bf16_t* vb = rk.vpool + ((e_page * rk.Hkv + g) * (size_t)rk.D + d) * 64 + vlm_vslot(e_within);   // [D][64 slots]
        vb[0] = f2bf(y0);
        vb[64] = f2bf(y1);
      }
    }
    return;
  }
  if (EPI & VLM_EPI_SWIGLU) {
#pragma unroll
    for (int r = 0; r < R; r += 2)
#pragma unroll
      for (int m = 0; m < MB; ++m)
        if (lane == (r >> 1) * MB + m && row[r] + 1 < N)
          y[(size_t)m * ldy + (row[r] >> 1)] = f2bf(swiglu_(rbf(acc[r][m]), rbf(acc[r + 1][m])));
    return;
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m)
      if (lane == r * MB + m && row[r] < N) {
        float v = acc[r][m];
        if (EPI & VLM_EPI_BIAS) v += bf2f(e_b0);
        if (EPI & VLM_EPI_RESIDUAL) v = rbf(v) + bf2f(e_r);
        y[(size_t)m * ldy + row[r]] = f2bf(v);
      }
}

// ------------------------------------------------------------------------------------------
// split-K kernel: K <= 2048 * NI; the 4 waves of a workgroup split K for RW rows
// ------------------------------------------------------------------------------------------
template <int RW, int NI, int MB, int EPI>
__global__ __launch_bounds__(256) void gemv_splitk_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                          const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                          bf16_t* __restrict__ y, int N, int K, int ldx, int ldw, int ldy,
                                                          int ldres) {
  __shared__ float red[4][RW][MB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = K >> 3;
  const int row0 = blockIdx.x * RW;
  // small latency-critical loads first (loads return in issue order): activations + epilogue operands
  u32x4_t xv[MB][NI];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ch = tid + 256 * i;
      xv[m][i] = *reinterpret_cast<const u32x4_t*>(x + (size_t)m * ldx + (size_t)min(ch, nchunk - 1) * 8);   // masked at use
    }
  bf16_t e_b = 0, e_r = 0;   // raw: no arithmetic before the weight loads are issued
  {
    const int tt = min(tid, RW * MB - 1), rr = min(row0 + tt / MB, N - 1);
    if (EPI & VLM_EPI_BIAS) e_b = bias[rr];
    if (EPI & VLM_EPI_RESIDUAL) e_r = res[(size_t)(tt % MB) * ldres + rr];
  }
  __builtin_amdgcn_sched_barrier(0);
  u32x4_t wv[RW][NI];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const bf16_t* wr = W + (size_t)min(row0 + r, N - 1) * ldw;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int ch = tid + 256 * i;
      wv[r][i] = ntl(wr + (size_t)min(ch, nchunk - 1) * 8);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  float acc[RW][MB];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const u32x4_t xm = (tid + 256 * i < nchunk) ? xv[m][i] : u32x4_t{0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < RW; ++r) acc[r][m] = dot8(wv[r][i], xm, acc[r][m]);
    }
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const float s = wave_sum(acc[r][m]);
      if (lane == 0) red[wave][r][m] = s;
    }
  __syncthreads();
  if (tid < RW * MB) {
    const int r = tid / MB, m = tid % MB;
    if (row0 + r < N) {
      float v = red[0][r][m] + red[1][r][m] + red[2][r][m] + red[3][r][m];
      if (EPI & VLM_EPI_BIAS) v += bf2f(e_b);
      if (EPI & VLM_EPI_RESIDUAL) v = rbf(v) + bf2f(e_r);
      y[(size_t)m * ldy + row0 + r] = f2bf(v);
    }
  }
}

// ------------------------------------------------------------------------------------------ launchers
struct Args {
  const void *x, *W, *bias, *res, *norm_w;
  void* y;
  int N, K, ldx, ldw, ldy, ldres;
  float eps;
  RopeKvArgs rk;
  AttnProArgs ap;
  hipStream_t st;
};

inline int launch_err() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

template <int R, int KC, int MB, int PRO, int EPI>
int launch_rw(const Args& a, int n_waves) {
  const int grid = vlm_cdiv(n_waves, 4);
  const size_t lds = (size_t)MB * a.K * 2 + (PRO == PRO_ATTN ? (size_t)MB * a.ap.Hq * a.ap.S * 2 * sizeof(float) : 0);
  hipLaunchKernelGGL((gemv_rowwave_kernel<R, KC, MB, PRO, EPI>), dim3(grid), dim3(256), lds, a.st, (const bf16_t*)a.x,
                     (const bf16_t*)a.W, (const bf16_t*)a.bias, (const bf16_t*)a.res, (const bf16_t*)a.norm_w,
                     (bf16_t*)a.y, a.N, a.K, a.ldx, a.ldw, a.ldy, a.ldres, a.eps, a.rk, a.ap);
  return launch_err();
}

template <int KC, int MB, int PRO, int EPI>
int launch_rw_r(const Args& a) {
  // R = 4 rows per wave once there are enough rows to give every CU several workgroups
  if (EPI != EPI_ROPE_KV && a.N >= 8192 && KC <= 3) return launch_rw<4, KC, MB, PRO, EPI>(a, vlm_cdiv(a.N, 4));
  if (EPI == EPI_ROPE_KV) {
    const int waves = (a.rk.Hq + a.rk.Hkv) * (a.rk.D / 2) + a.rk.Hkv * a.rk.D / 2;
    return launch_rw<2, KC, MB, PRO, EPI>(a, waves);
  }
  return launch_rw<2, KC, MB, PRO, EPI>(a, vlm_cdiv(a.N, 2));
}

template <int MB, int PRO, int EPI>
int launch_rw_k(const Args& a) {
  if (a.K <= 512 * 3) return launch_rw_r<3, MB, PRO, EPI>(a);
  if (a.K <= 512 * 7) return launch_rw_r<7, MB, PRO, EPI>(a);
  if constexpr (PRO == PRO_RMSNORM)      // hidden 4096 (Mistral-7B / Idefics2-8B): the norm-prologue forms only
    if (a.K <= 512 * 8) return launch_rw_r<8, MB, PRO, EPI>(a);
  return VLM_ERR_SHAPE;
}

template <int PRO, int EPI>
int launch_rw_m(int M, const Args& a) {
  switch (M) {
    case 1: return launch_rw_k<1, PRO, EPI>(a);
    case 2: return launch_rw_k<2, PRO, EPI>(a);
    case 4: return launch_rw_k<4, PRO, EPI>(a);
    case 8: return launch_rw_k<8, PRO, EPI>(a);
    default: return VLM_ERR_SHAPE;
  }
}

template <int RW, int NI, int MB, int EPI>
int launch_sk(const Args& a) {
  hipLaunchKernelGGL((gemv_splitk_kernel<RW, NI, MB, EPI>), dim3(vlm_cdiv(a.N, RW)), dim3(256), 0, a.st, (const bf16_t*)a.x,
                     (const bf16_t*)a.W, (const bf16_t*)a.bias, (const bf16_t*)a.res, (bf16_t*)a.y, a.N, a.K, a.ldx, a.ldw,
                     a.ldy, a.ldres);
  return launch_err();
}

int g_gemv_variant = 0;   // A/B bits (VLM_TUNE_GEMV_VARIANT)

template <int MB, int EPI>
int launch_sk_k(const Args& a) {
  const int ni = vlm_cdiv(a.K, 2048);
  if (MB <= 2) {
    // N = 1536 rows over 256 CUs: 4 rows per workgroup = 384 workgroups = 1.5 per CU (half the CUs stream 143 KB, half
    // 72 KB: the launch ends with the loaded half); 6 rows = ONE 107 KB workgroup per CU, 3 rows = two per CU
    if constexpr (MB == 1) {
      if (ni <= 5 && (g_gemv_variant & 1) && a.N % 6 == 0) return launch_sk<6, 5, MB, EPI>(a);
      if (ni <= 5 && (g_gemv_variant & 2) && a.N % 3 == 0) return launch_sk<3, 5, MB, EPI>(a);
    }
    if (ni <= 5) return launch_sk<4, 5, MB, EPI>(a);
    if (ni <= 10) return launch_sk<2, 10, MB, EPI>(a);
  } else {
    if (ni <= 5) return launch_sk<2, 5, MB, EPI>(a);
    if (ni <= 10) return launch_sk<1, 10, MB, EPI>(a);
  }
  return VLM_ERR_SHAPE;
}

template <int EPI>
int launch_sk_m(int M, const Args& a) {
  switch (M) {
    case 1: return launch_sk_k<1, EPI>(a);
    case 2: return launch_sk_k<2, EPI>(a);
    case 4: return launch_sk_k<4, EPI>(a);
    case 8: return launch_sk_k<8, EPI>(a);
    default: return VLM_ERR_SHAPE;
  }
}

}  // namespace

VLM_INTERNAL void vlm_gemv_set_variant(int bits) { g_gemv_variant = bits; }

extern "C" int vlm_gemv_bf16(const void* x, const void* W, const void* bias, const void* res, const void* norm_w,
                             void* y, int M, int N, int K, int ldx, int ldw, int ldy, int ldres, float eps,
                             int epilogue, void* stream) {
  return vlm_gemv_bf16_ex(x, W, bias, res, norm_w, y, M, N, K, ldx, ldw, ldy, ldres, eps, epilogue, 1, nullptr, stream);
}

extern "C" size_t vlm_gemv_workspace_bytes(void) { return vlm_gemv_mfma_ws_bytes(); }

extern "C" int vlm_gemv_bf16_ws(const void* x, const void* W, const void* bias, const void* res, const void* norm_w, void* y,
                                int M, int N, int K, int ldx, int ldw, int ldy, int ldres, float eps, int epilogue,
                                void* workspace, void* stream) {
  return vlm_gemv_bf16_ex(x, W, bias, res, norm_w, y, M, N, K, ldx, ldw, ldy, ldres, eps, epilogue, 1, workspace, stream);
}

extern "C" int vlm_gemv_qkv_rope_kvwrite_ws(const void* h, const void* norm_w, float eps, const void* Wqkv, const void* bqkv,
                                            void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D, const void* pos,
                                            const void* slot, const void* inv_freq, const void* block_table, int max_pages,
                                            void* kpool, void* vpool, void* workspace, void* stream) {
  return vlm_gemv_qkv_rope_kvwrite_ex(h, norm_w, eps, Wqkv, bqkv, qkv, ldq, M, hidden, Hq, Hkv, D, pos, slot, inv_freq, block_table,
                                      max_pages, kpool, vpool, 1, workspace, 1.f, 0, stream);
}

// mfma: 0 = v_dot2c kernels only; ws: the engine's workspace for vlm_gemv_mfma_try (nullptr: no K split over workgroups)
VLM_INTERNAL int vlm_gemv_bf16_ex(const void* x, const void* W, const void* bias, const void* res, const void* norm_w,
                                  void* y, int M, int N, int K, int ldx, int ldw, int ldy, int ldres, float eps,
                                  int epilogue, int mfma, void* ws, void* stream) {
  if (!x || !W || !y || N <= 0 || K <= 0) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_BIAS) && !bias) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_RESIDUAL) && !res) return VLM_ERR_ARG;
  if (K % 8 != 0 || ldx % 8 != 0 || ldw % 8 != 0) return VLM_ERR_SHAPE;
  if ((epilogue & VLM_EPI_SWIGLU) && (N % 2 != 0)) return VLM_ERR_SHAPE;
  // activations in the tiled layout between two projections of a batched decode step (include/vlm_hip.h VLM_EPI_X_TILED / Y_TILED):
  // only the matrix-core forms know it - a shape they do not take is refused, nothing is enqueued
  if (epilogue & VLM_EPI_X_TILED) {
    if (!mfma || norm_w || (epilogue & VLM_EPI_Y_TILED)) return VLM_ERR_SHAPE;
    const int rc = vlm_gemv_mfma_rows_try(x, W, bias, res, y, M, N, K, ldw, ldy, ldres, epilogue & ~VLM_EPI_X_TILED, stream);
    return rc >= 0 ? rc : VLM_ERR_SHAPE;
  }
  if (mfma) {   // (M >= 3 by default) batch rows as the N dimension of the matrix cores (gemv_mfma.hip); -1: shape not handled there
    const int rc = vlm_gemv_mfma_try(x, W, bias, res, norm_w, y, M, N, K, ldx, ldw, ldy, ldres, eps, epilogue, nullptr, ws, stream);
    if (rc >= 0) return rc;
  }
  if (epilogue & VLM_EPI_Y_TILED) return VLM_ERR_SHAPE;
  if ((size_t)M * K * 2 > 64 * 1024 && (norm_w || K <= 3584)) return VLM_ERR_SHAPE;
  Args a{x, W, bias, res, norm_w, y, N, K, ldx, ldw, ldy, ldres, eps, RopeKvArgs{}, AttnProArgs{}, (hipStream_t)stream};
  if (K <= 3584 || (norm_w && K <= 4096)) {
#define GO(P, E) return launch_rw_m<P, E>(M, a)
    if (norm_w) {
      switch (epilogue) {
        case VLM_EPI_NONE: GO(PRO_RMSNORM, VLM_EPI_NONE);
        case VLM_EPI_BIAS: GO(PRO_RMSNORM, VLM_EPI_BIAS);
        case VLM_EPI_SWIGLU: GO(PRO_RMSNORM, VLM_EPI_SWIGLU);
        default: return VLM_ERR_ARG;
      }
    }
    switch (epilogue) {
      case VLM_EPI_NONE: GO(PRO_NONE, VLM_EPI_NONE);
      case VLM_EPI_BIAS: GO(PRO_NONE, VLM_EPI_BIAS);
      case VLM_EPI_RESIDUAL: GO(PRO_NONE, VLM_EPI_RESIDUAL);
      case VLM_EPI_SWIGLU: GO(PRO_NONE, VLM_EPI_SWIGLU);
      default: return VLM_ERR_ARG;
    }
#undef GO
  }
  if (norm_w) return VLM_ERR_SHAPE;   // the norm prologue needs the row-wave kernel (K <= 3584)
  switch (epilogue) {
    case VLM_EPI_NONE: return launch_sk_m<VLM_EPI_NONE>(M, a);
    case VLM_EPI_BIAS: return launch_sk_m<VLM_EPI_BIAS>(M, a);
    case VLM_EPI_RESIDUAL: return launch_sk_m<VLM_EPI_RESIDUAL>(M, a);
    default: return VLM_ERR_ARG;
  }
}

extern "C" int vlm_gemv_qkv_rope_kvwrite(const void* h, const void* norm_w, float eps, const void* Wqkv,
                                         const void* bqkv, void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D,
                                         const void* pos, const void* slot, const void* inv_freq,
                                         const void* block_table, int max_pages, void* kpool, void* vpool,
                                         void* stream) {
  return vlm_gemv_qkv_rope_kvwrite_ex(h, norm_w, eps, Wqkv, bqkv, qkv, ldq, M, hidden, Hq, Hkv, D, pos, slot, inv_freq, block_table,
                                      max_pages, kpool, vpool, 1, nullptr, 1.f, 0, stream);
}

VLM_INTERNAL int vlm_gemv_qkv_rope_kvwrite_ex(const void* h, const void* norm_w, float eps, const void* Wqkv,
                                              const void* bqkv, void* qkv, int ldq, int M, int hidden, int Hq, int Hkv, int D,
                                              const void* pos, const void* slot, const void* inv_freq,
                                              const void* block_table, int max_pages, void* kpool, void* vpool, int mfma,
                                              void* ws, float qk_scale, int long_from, void* stream) {
  if (!h || !norm_w || !Wqkv || !bqkv || !qkv || !pos || !slot || !inv_freq || !kpool || !vpool || max_pages <= 0)
    return VLM_ERR_ARG;
  const int N = (Hq + 2 * Hkv) * D;
  if (mfma) {
    const VlmRopeKv rk{(const int*)pos, (const int*)slot, (const float*)inv_freq, (const int*)block_table, max_pages, Hq, Hkv, D,
                       (unsigned short*)kpool, (unsigned short*)vpool, qk_scale, long_from};
    const int rc = vlm_gemv_mfma_try(h, Wqkv, bqkv, nullptr, norm_w, qkv, M, N, hidden, hidden, hidden, ldq, 0, eps, VLM_EPI_BIAS, &rk,
                                     ws, stream);
    if (rc >= 0) return rc;
  }
  if (hidden % 8 || D % 16 || hidden > 4096 || (size_t)M * hidden * 2 > 64 * 1024) return VLM_ERR_SHAPE;
  Args a{h, Wqkv, bqkv, nullptr, norm_w, qkv, N, hidden, hidden, hidden, ldq, 0, eps,
         RopeKvArgs{(const int*)pos, (const int*)slot, (const float*)inv_freq, (const int*)block_table, max_pages, Hq, Hkv,
                    D, (bf16_t*)kpool, (bf16_t*)vpool, qk_scale, long_from},
         AttnProArgs{}, (hipStream_t)stream};
  return launch_rw_m<PRO_RMSNORM, EPI_ROPE_KV>(M, a);
}

extern "C" int vlm_gemv_attn_out_bf16(const void* part_o, const void* part_ml, int nsplit, const void* Wo, void* h, int ldh,
                                      int N, int Hq, int D, void* stream) {
  if (!part_o || !part_ml || !Wo || !h || nsplit <= 0 || nsplit > AT2_S) return VLM_ERR_ARG;
  const int K = Hq * D;
  if (K % 8 || K > 2048 || D % 8) return VLM_ERR_SHAPE;     // one x chunk per thread
  Args a{nullptr, Wo, nullptr, h, nullptr, h, N, K, K, K, ldh, ldh, 0.f, RopeKvArgs{},
         AttnProArgs{(const float*)part_o, (const float*)part_ml, nsplit, Hq, D}, (hipStream_t)stream};
  return launch_rw_k<1, PRO_ATTN_PS, VLM_EPI_RESIDUAL>(a);
}

extern "C" int vlm_gemv_attn_out(const void* part_o, const void* part_ml, int nsplit, const void* Wo, void* h, int ldh,
                                 int M, int N, int Hq, int D, void* stream) {
  if (!part_o || !part_ml || !Wo || !h || nsplit <= 0) return VLM_ERR_ARG;
  const int K = Hq * D;
  if (K % 8 || K > 3584 || (size_t)M * K * 2 > 64 * 1024) return VLM_ERR_SHAPE;
  Args a{nullptr, Wo, nullptr, h, nullptr, h, N, K, K, K, ldh, ldh, 0.f, RopeKvArgs{},
         AttnProArgs{(const float*)part_o, (const float*)part_ml, nsplit, Hq, D}, (hipStream_t)stream};
  return launch_rw_m<PRO_ATTN, VLM_EPI_RESIDUAL>(M, a);
}
