// The fused decode block of a ONE-row step on MI355X: [page-split attention] -> [split merge + o_proj + residual] ->
// [RMSNorm + gate/up GEMV + SwiGLU] as ONE launch instead of three.
//
// Replaces, for a one-token call (behaviour, not code):
//   mx.fast.scaled_dot_product_attention at L == 1     reference mlx_vlm/models/base.py:366-373
//   o_proj + residual, post_attention_layernorm         reference mlx_vlm/models/qwen2_vl/language.py:115-120,149-153
//   gate_proj / up_proj + SwiGLU                        reference mlx_vlm/models/qwen2_vl/language.py:123-133 (mlp), activations.py:7-9
//
// Why (profiles/r03_bench_kernel_stats.txt, r04_attn_block_probe.txt): at Qwen2-VL-2B dims the attention launch (0.46 MB)
// and the o_proj launch (4.7 MB) are dependent-latency chains of 4.7 us each during which the memory system idles, and the
// gate/up launch (55 MB, 10.4 us) cannot start its weight stream before they end.  Here the 55 MB are REQUESTED AT ENTRY:
// every workgroup's seven gate/up waves issue all their weight loads (10 rows x 3 chunks x 16 B per lane, register
// resident - 215 KB per CU, 42 % of its VGPR file) and park at the workgroup barrier; the eighth wave of each workgroup
// runs the latency chain meanwhile:
//   workgroups 0 .. Hkv*S-1   one page-split attention unit (kv head g, page stride s): the page walk of attn_decode.hip,
//                             partials out with write-through (sc1) stores + drained flag (Guideline 16, R1);
//                hop 2a       the S units of a kv head wait for each other's flags, each merges ITS G*16/S chunks of the
//                             attention output over all splits (the arithmetic of the o_proj prologue it replaces:
//                             vlm_merge_splits16) and publishes them as 8-byte {data, tag} granules;
//   the other workgroups      o_proj rows (768 row pairs over 224 waves: 3 or 4 pairs each, weights loaded at entry);
//                hop 2b       gather the 768 granules of the merged vector (tags checked, retried), dot, + residual, publish
//                             the new residual stream as granules (and plain stores for the down projection's launch);
//   every workgroup, hop 3    its eighth wave gathers the residual stream, RMSNorm -> LDS -> barrier; the gate/up waves
//                             dot their register-resident rows with it, SwiGLU, store.
// Hand-offs are MI355X_MICROARCH.md's granules (one naturally aligned 8-byte {data, tag} per sc1 store, tag = a launch
// epoch, no separate flag, no fence; swept with sc1 loads until every tag matches) except the partials (24 KB: R1 form).
// The epoch is a device word advanced by ONE thread of the preceding qkv launch (gemv_bf16.hip, RopeKvArgs.epoch) - a
// replayed hipGraph therefore tags every launch differently with no host work - or by a one-thread kernel (bump_epoch).
// ALL 256 workgroups of 512 threads are co-resident (one per CU: checked against the occupancy query and the CU count at
// first use; the entry point refuses otherwise), which is what makes waiting on another workgroup legal; every wait is
// bounded and raises the error word instead of hanging.
// Numerics: every value is produced by the same source expressions as the three launches (page walk, split merge, v_dot2c
// row dots in the same chunk order, DPP wave sums, RMSNorm, SwiGLU rounding points): results are bit-identical to them
// (tests/test_decode_block_gpu.py).
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include "common.cuh"
#include "attn_pagesplit.cuh"
#include "internal.h"
#include "../../include/vlm_hip.h"

namespace {

typedef unsigned long long u64;
constexpr int HD = VLM_HD;
constexpr int BLK_WGS = 256, BLK_THREADS = 512, GU_WAVES = 7;
constexpr int RP = 4;                       // o_proj row pairs per wave, at most
constexpr int RG0 = 10, RG1 = 12;           // gate / up rows per wave: LAYOUT 0 (1792 waves) / LAYOUT 1 (7 (256 - NU) waves)
constexpr int POLL_LIMIT = 1 << 18;         // bounded waits (a fraction of a second): then the error word, never a hang

// workspace (bytes): epoch word | error word | flags of the attention units | granules of the merged attention vector |
// granules of the new residual stream | debug stamps
constexpr size_t WS_EPOCH = 0, WS_ERR = 4, WS_F1 = 256, WS_XM = 1024, WS_HG = 1024 + 8192, WS_STAMPS = 1024 + 16384,
                 WS_BYTES = 32768;

struct BlockArgs {
  const bf16_t* q;            // [Hq][HD] of the step's row (the qkv launch's output)
  const bf16_t* kpool;
  const bf16_t* vpool;
  const int* block_table;
  const int* kv_len;
  int ldq, max_pages, Hkv, kv_len_add, S;
  float scale_log2;
  bf16_t* part_o;             // [Hq][S][HD] bf16
  float* part_ml;             // [Hq][S][2]
  const bf16_t* wo;           // [K][K]
  bf16_t* h;                  // [K] residual stream, updated in place (read by the down projection's launch)
  const bf16_t* ln2_w;        // [K]
  const bf16_t* wgu;          // [n2][K] gate / up rows interleaved
  bf16_t* act;                // [n2 / 2]
  int n2;
  float eps;
  char* ws;
  u64* stamps;                // nullptr, or [16] wall-clock ticks (debug timeline)
};

__device__ __forceinline__ u32x4_t ntl(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); }
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
// agent-scope (sc1) accesses: L2-served loads that bypass the CU's L1, write-through stores
__device__ __forceinline__ u64 ld8(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st8(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld4(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st4(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define VLM_RSRC(ptr) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, 0x7fffffff, 0x00020000)
#define VLM_LD16(rs, off) __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off), 0, 16)        /* aux 16 = sc1 */
#define VLM_ST16(v, rs, off) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(off), 0, 16)

// Gather a K-element bf16 vector published as K / 2 granules {2 x bf16, tag}: lane l takes the chunks l, l + 64, ... (8
// elements = 4 granules = two 16-byte sc1 loads each), every tag checked, the sweep repeated until all match.  A cheap
// pre-poll (ONE 8-byte load per lane over a spread sample, with s_sleep) keeps the full sweeps off the memory pipe while
// the producers are still far away (MI355X_MICROARCH.md: polling-cost).
template <int KC, bool PREPOLL, typename RS>
__device__ __forceinline__ void gather_vec(RS rs, const u64* base, unsigned E, int lane, u32x4_t (&x)[KC], unsigned* err) {
  int it = 0;
  if (PREPOLL) {
    for (; it < POLL_LIMIT; ++it) {
      const u64 v = ld8(base + lane * (KC * 4));
      if (__all((unsigned)(v >> 32) == E)) break;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  for (; it < POLL_LIMIT; ++it) {
    u32x4_t a[KC], b[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      a[c] = VLM_LD16(rs, (lane + 64 * c) * 32);
      b[c] = VLM_LD16(rs, (lane + 64 * c) * 32 + 16);
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      ok = ok && a[c][1] == E && a[c][3] == E && b[c][1] == E && b[c][3] == E;
      x[c] = u32x4_t{a[c][0], a[c][2], b[c][0], b[c][2]};
    }
    if (__all(ok)) return;
    __builtin_amdgcn_s_sleep(1);
  }
  if (lane == 0) atomicAdd(err, 1u);
}

// A wave's weight rows, REQUESTED IN A PACED STREAM: at most DEPTH 1-KiB loads of the wave in flight (s_waitcnt vmcnt(DEPTH - 1)
// before every further request).  Everything-at-entry (DEPTH >= the wave's loads) puts 60 MB into the memory system's queues
// in the first two microseconds - a chain hop issued after that waits behind all of it (profiles/r04_decode_block_v1.txt);
// 7 waves x DEPTH KiB per CU in flight is bandwidth x latency, the queues stay short and the stream runs at the same rate.
template <int N, int KC, int DEPTH>
__device__ __forceinline__ void paced_rows(const bf16_t* W, int row0, int row_max, int K, int lane, u32x4_t (&wv)[N][KC]) {
#pragma unroll
  for (int i = 0; i < N * KC; ++i) {
    const int r = i / KC, c = i % KC;
    if (DEPTH < N * KC && i >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    wv[r][c] = ntl(W + (size_t)min(row0 + r, row_max) * K + (size_t)(lane + 64 * c) * 8);
  }
}

struct Sync {                 // the hand-off state of a launch, decoded from the workspace
  unsigned E;
  unsigned* err;
  unsigned* f1;
  u64* xm;
  u64* hg;
};
__device__ __forceinline__ Sync sync_of(const BlockArgs& p) {
  return Sync{*reinterpret_cast<const unsigned*>(p.ws + WS_EPOCH), reinterpret_cast<unsigned*>(p.ws + WS_ERR),
              reinterpret_cast<unsigned*>(p.ws + WS_F1), reinterpret_cast<u64*>(p.ws + WS_XM), reinterpret_cast<u64*>(p.ws + WS_HG)};
}

// ---- gate / up rows row0 .. row0 + RG - 1 of one wave: every weight load at entry, then the workgroup barrier behind which
//      the normalised residual stream is in LDS, dots, SwiGLU
template <int KC, int RG, int DEPTH>
__device__ __forceinline__ void gateup_rows(const BlockArgs& p, const uint4* xs, int row0, int lane, u64* st) {
  constexpr int K = KC * 512;
  u32x4_t wv[RG][KC];
  paced_rows<RG, KC, DEPTH>(p.wgu, row0, p.n2 - 1, K, lane, wv);
  __builtin_amdgcn_sched_barrier(0);
  if (st) st[8] = wall_clock64();                                                  // weights requested
  asm volatile("s_barrier" ::: "memory");                                          // the eighth wave has x in LDS
  if (st) st[9] = wall_clock64();
  float acc[RG];
#pragma unroll
  for (int r = 0; r < RG; ++r) acc[r] = 0.f;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(&xs[lane + 64 * c]);
#pragma unroll
    for (int r = 0; r < RG; ++r) acc[r] = dot8(wv[r][c], xv, acc[r]);
  }
#pragma unroll
  for (int r = 0; r < RG; ++r) acc[r] = wave_sum(acc[r]);
#pragma unroll
  for (int r = 0; r < RG; r += 2)
    if (lane == (r >> 1) && row0 + r + 1 < p.n2) p.act[(row0 + r) >> 1] = f2bf(swiglu_(rbf(acc[r]), rbf(acc[r + 1])));
  if (st) st[10] = wall_clock64();
}

// ---- attention unit (kv head g, page stride s): page walk, partials out (R1 form), hop 2a, its slice of the merged vector
template <int G, bool IDENT>
__device__ __forceinline__ void attention_unit(const BlockArgs& p, const Sync& y, int unit, int lane, u64* st) {
  const int g = unit / p.S, s = unit % p.S;
  f32x4_t ot[8];
  float m_run, l_run;
  int npages;
  vlm_pagesplit_walk<G, IDENT>(p.q, p.kpool, p.vpool, p.block_table, p.kv_len, p.ldq, p.max_pages, p.Hkv, p.kv_len_add,
                               p.scale_log2, p.S, 0, g, s, lane, ot, m_run, l_run, npages);
  const int head = lane & 15, gq = lane >> 4;
  l_run = col4_sum(l_run);
  // partials in the layout of the unfused attention launch (pagesplit_finish, MERGE = false), write-through
  if (head < G) {
    const size_t e = (size_t)(g * G + head) * p.S + s;
    if (gq == 0) st8(reinterpret_cast<u64*>(p.part_ml + e * 2), ((u64)__float_as_uint(l_run) << 32) | __float_as_uint(m_run));
    if (m_run != -INFINITY) {
      bf16_t* po = p.part_o + e * HD + 4 * gq;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt)
        st8(reinterpret_cast<u64*>(po + 16 * dt), ((u64)pack_bf2(ot[dt][2], ot[dt][3]) << 32) | pack_bf2(ot[dt][0], ot[dt][1]));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 // every store has left (R1) ...
  if (lane == 0) st4(y.f1 + unit, y.E);                                            // ... then the flag
  if (st) st[1] = wall_clock64();                                                  // partial published
  // hop 2a: the S units of this kv head
  {
    int it = 0;
    for (; it < POLL_LIMIT; ++it) {
      const unsigned v = ld4(y.f1 + g * p.S + min(lane, p.S - 1));
      if (__all(v == y.E)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    if (it == POLL_LIMIT && lane == 0) atomicAdd(y.err, 1u);
  }
  if (st) st[2] = wall_clock64();                                                  // all partials of the kv head are out
  // my chunks of the attention output: chunk cj of kv head g = (head g*G + cj / 16, d0 = 8 (cj % 16)); lane j < cpu one each
  const int cpu = G * 16 / p.S, cj = s * cpu + min(lane, cpu - 1);
  const int mh = g * G + (cj >> 4), d0 = (cj & 15) * 8;
  const auto rs_po = VLM_RSRC(p.part_o);
  const auto rs_xm = VLM_RSRC(y.xm);
  float2 a2_ml[VLM_MERGE_S];
  u32x4_t a2_o[VLM_MERGE_S];
#pragma unroll
  for (int sp = 0; sp < VLM_MERGE_S; ++sp) {
    const int e = mh * p.S + min(sp, p.S - 1);
    const u64 mlv = ld8(reinterpret_cast<const u64*>(p.part_ml) + e);
    a2_ml[sp] = make_float2(__uint_as_float((unsigned)mlv), __uint_as_float((unsigned)(mlv >> 32)));
    a2_o[sp] = VLM_LD16(rs_po, (e * HD + d0) * 2);
  }
  const uint4 o = vlm_merge_splits16(a2_ml, a2_o, p.S);
  if (lane < cpu) {
    const int off = (mh * HD + d0) * 4;                                            // 4 granules of 8 bytes per chunk
    VLM_ST16((u32x4_t{o.x, y.E, o.y, y.E}), rs_xm, off);
    VLM_ST16((u32x4_t{o.z, y.E, o.w, y.E}), rs_xm, off + 16);
  }
  if (st) st[3] = wall_clock64();                                                  // merged slice published
}

// ---- o_proj row pairs of wave wi of NO: weights at entry, hop 2b, dots, + residual, publish
template <int KC, int DEPTH>
__device__ __forceinline__ void oproj_rows(const BlockArgs& p, const Sync& y, int wi, int NO, int lane, u64* st) {
  constexpr int K = KC * 512;
  const int pairs = K / 2, base = pairs / NO, rem = pairs % NO;
  const int cnt = base + (wi < rem ? 1 : 0), p0 = wi * base + min(wi, rem);
  const unsigned res2 = reinterpret_cast<const unsigned*>(p.h)[min(p0 + lane, pairs - 1)];     // rows 2 (p0 + lane), + 1
  __builtin_amdgcn_sched_barrier(0);
  u32x4_t wo[2 * RP][KC];
  paced_rows<2 * RP, KC, DEPTH>(p.wo, 2 * p0, K - 1, K, lane, wo);
  __builtin_amdgcn_sched_barrier(0);
  // hop 2b: the merged attention vector
  u32x4_t xa[KC];
  gather_vec<KC, true>(VLM_RSRC(y.xm), y.xm, y.E, lane, xa, y.err);
  if (st) st[5] = wall_clock64();                                                  // attention output here
  float acc[2 * RP];
#pragma unroll
  for (int r = 0; r < 2 * RP; ++r) acc[r] = 0.f;
#pragma unroll
  for (int c = 0; c < KC; ++c)
#pragma unroll
    for (int r = 0; r < 2 * RP; ++r) acc[r] = dot8(wo[r][c], xa[c], acc[r]);
#pragma unroll
  for (int r = 0; r < 2 * RP; ++r) acc[r] = wave_sum(acc[r]);
  // lane j < cnt: row pair p0 + j.  h = bf16(bf16(o_proj) + residual) as the o_proj launch's epilogue rounds it
  float a0 = acc[0], a1 = acc[1];
#pragma unroll
  for (int j = 1; j < RP; ++j) {
    a0 = (lane == j) ? acc[2 * j] : a0;
    a1 = (lane == j) ? acc[2 * j + 1] : a1;
  }
  const unsigned payload = (unsigned)f2bf(rbf(a0) + bf_lo(res2)) | ((unsigned)f2bf(rbf(a1) + bf_hi(res2)) << 16);
  if (lane < cnt) {
    st8(y.hg + p0 + lane, ((u64)y.E << 32) | payload);
    reinterpret_cast<unsigned*>(p.h)[p0 + lane] = payload;
  }
  if (st) st[6] = wall_clock64();                                                  // residual rows published
}

// ---- hop 3: gather the new residual stream, RMSNorm (the rounding points of the gate/up launch's prologue,
//      gemv_bf16.hip PRO_RMSNORM) -> LDS -> the workgroup barrier the gate/up waves wait at
template <int KC, bool PREPOLL>
__device__ __forceinline__ void norm_to_lds(const BlockArgs& p, const Sync& y, const uint4 (&nwv)[KC], uint4* xs, int lane, u64* st,
                                            int st_slot) {
  constexpr int K = KC * 512;
  u32x4_t hv[KC];
  gather_vec<KC, PREPOLL>(VLM_RSRC(y.hg), y.hg, y.E, lane, hv, y.err);
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const float v[8] = {bf_lo(hv[c][0]), bf_hi(hv[c][0]), bf_lo(hv[c][1]), bf_hi(hv[c][1]),
                        bf_lo(hv[c][2]), bf_hi(hv[c][2]), bf_lo(hv[c][3]), bf_hi(hv[c][3])};
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
  }
  const float inv = rsqrtf(wave_sum(ss) / (float)K + p.eps);
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    const u32x4_t u = hv[c];
    const uint4 wu = nwv[c];
    uint4 o;
    o.x = pack_bf2(bf_lo(wu.x) * rbf(bf_lo(u[0]) * inv), bf_hi(wu.x) * rbf(bf_hi(u[0]) * inv));
    o.y = pack_bf2(bf_lo(wu.y) * rbf(bf_lo(u[1]) * inv), bf_hi(wu.y) * rbf(bf_hi(u[1]) * inv));
    o.z = pack_bf2(bf_lo(wu.z) * rbf(bf_lo(u[2]) * inv), bf_hi(wu.z) * rbf(bf_hi(u[2]) * inv));
    o.w = pack_bf2(bf_lo(wu.w) * rbf(bf_lo(u[3]) * inv), bf_hi(wu.w) * rbf(bf_hi(u[3]) * inv));
    xs[lane + 64 * c] = o;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (st) st[st_slot] = wall_clock64();                                            // normalised stream in LDS
  asm volatile("s_barrier" ::: "memory");
}

// LAYOUT 1 (the product): the latency chain runs on CUs of its OWN - workgroups 0 .. NU-1 hold one attention unit (eighth
// wave) and the o_proj rows (waves 0..6: 768 row pairs over 7 NU waves), nothing else, so their loads never queue behind a
// weight stream; workgroups NU .. 255 hold the gate/up rows (RG1 per wave, requested at entry) and an eighth wave that
// gathers the residual stream (first sweep issued right away: it returns when the CU's weight requests have drained).
// LAYOUT 0 (first cut, kept for the A/B that explains LAYOUT 1: profiles/r04_decode_block_v1.txt): every workgroup = 7
// gate/up waves + one chain wave on the same CU.
template <int G, bool IDENT, int KC, int LAYOUT, int DEPTH>
__global__ __launch_bounds__(BLK_THREADS) void decode_block_kernel(BlockArgs p) {
  __shared__ __attribute__((aligned(16))) uint4 xs[KC * 64];      // the normalised residual stream, bf16 [K]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wg = blockIdx.x;
  const int NU = p.Hkv * p.S;
  u64* const st0 = (p.stamps && wg == 0 && lane == 0) ? p.stamps : nullptr;                    // an attention unit's workgroup
  u64* const st1 = (p.stamps && wg == (LAYOUT ? NU - 1 : BLK_WGS - 1) && wave == (LAYOUT ? 0 : 7) && lane == 0) ? p.stamps : nullptr;  // o_proj rows
  u64* const st2 = (p.stamps && wg == BLK_WGS - 1 && lane == 0) ? p.stamps : nullptr;          // gate / up rows + hop 3

  if (LAYOUT == 0) {
    if (wave < GU_WAVES) {
      gateup_rows<KC, RG0, DEPTH>(p, xs, (wg * GU_WAVES + wave) * RG0, lane, wave == 0 ? st2 : nullptr);
      return;
    }
    const Sync y = sync_of(p);
    uint4 nwv[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) nwv[c] = reinterpret_cast<const uint4*>(p.ln2_w)[lane + 64 * c];
    if (st0) st0[0] = wall_clock64();
    if (st1) st1[4] = wall_clock64();
    if (wg < NU) attention_unit<G, IDENT>(p, y, wg, lane, st0);
    else oproj_rows<KC, DEPTH>(p, y, wg - NU, BLK_WGS - NU, lane, st1);
    norm_to_lds<KC, true>(p, y, nwv, xs, lane, wg == 0 ? st0 : st2, wg == 0 ? 11 : 7);
    return;
  }
  if (wg < NU) {                                                   // ---- chain workgroups
    const Sync y = sync_of(p);
    if (wave == 7) {
      if (st0) st0[0] = wall_clock64();
      attention_unit<G, IDENT>(p, y, wg, lane, st0);
    } else {
      if (st1) st1[4] = wall_clock64();
      oproj_rows<KC, DEPTH>(p, y, wg * GU_WAVES + wave, NU * GU_WAVES, lane, st1);
    }
    return;
  }
  if (wave < GU_WAVES) {                                           // ---- streaming workgroups
    gateup_rows<KC, RG1, DEPTH>(p, xs, ((wg - NU) * GU_WAVES + wave) * RG1, lane, wave == 0 ? st2 : nullptr);
    return;
  }
  const Sync y = sync_of(p);
  uint4 nwv[KC];
#pragma unroll
  for (int c = 0; c < KC; ++c) nwv[c] = reinterpret_cast<const uint4*>(p.ln2_w)[lane + 64 * c];
  norm_to_lds<KC, false>(p, y, nwv, xs, lane, st2, 7);
}

__global__ void epoch_bump_kernel(unsigned* e) { *e += 1u; }

// co-residency of the whole grid: one 512-thread workgroup per CU on a part with at least BLK_WGS CUs (checked once)
template <typename Kern>
bool resident(Kern kern) {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, BLK_THREADS, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
  return cus >= BLK_WGS && per_cu >= 1;
}

constexpr int KC_ = 3;                    // hidden = Hq * 128 = 1536
int g_layout = 1;                         // VLM_DECODE_BLOCK_LAYOUT=0: the first cut (A/B)
int g_depth = 8;                          // VLM_DECODE_BLOCK_DEPTH: loads in flight per wave (A/B: 4, 8, 16, 64 = all at entry)

bool shape_ok(int Hq, int Hkv, int D, int inter, int nsplit) {
  if (D != HD || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0 || inter <= 0 || nsplit <= 0 || nsplit > VLM_MERGE_S) return false;
  const int K = Hq * HD, G = Hq / Hkv, NU = Hkv * nsplit;
  if (K != KC_ * 512) return false;
  if (G != 3 && G != 6) return false;
  if ((G * 16) % nsplit != 0 || NU > 32) return false;
  if (g_layout == 0) {
    if (2 * inter > BLK_WGS * GU_WAVES * RG0) return false;
    const int NO = BLK_WGS - NU, pairs = K / 2;
    return (pairs + NO - 1) / NO <= RP;
  }
  if (2 * inter > (BLK_WGS - NU) * GU_WAVES * RG1) return false;
  const int NO = NU * GU_WAVES, pairs = K / 2;
  return (pairs + NO - 1) / NO <= RP;
}

}  // namespace

extern "C" size_t vlm_decode_block_ws_bytes(void) { return WS_BYTES; }

extern "C" int vlm_decode_block_supported(int Hq, int Hkv, int D, int inter, int nsplit) {
  static int res = -1;                      // (the answer depends on the device and the kernel's registers, not on the shape)
  if (res < 0) {
    const char* e = getenv("VLM_DECODE_BLOCK_LAYOUT");
    if (e) g_layout = atoi(e) ? 1 : 0;
    e = getenv("VLM_DECODE_BLOCK_DEPTH");
    if (e) g_depth = atoi(e);
    res = resident(decode_block_kernel<6, true, KC_, 1, 8>) && resident(decode_block_kernel<6, false, KC_, 1, 8>) &&
                  resident(decode_block_kernel<6, true, KC_, 1, 64>) && resident(decode_block_kernel<6, true, KC_, 0, 64>) ? 1 : 0;
  }
  if (!shape_ok(Hq, Hkv, D, inter, nsplit)) return 0;
  return res;
}

extern "C" int vlm_decode_block_bf16(const void* q, int ldq, const void* kpool, const void* vpool, const void* block_table,
                                     int max_pages, const void* kv_len, int kv_len_add, int Hq, int Hkv, int D, float scale,
                                     int nsplit, void* part_o, void* part_ml, const void* Wo, void* h, const void* ln2_w,
                                     float eps, const void* Wgu, int inter, void* act, void* ws, int bump_epoch, void* stream) {
  if (!q || !kpool || !vpool || !kv_len || !part_o || !part_ml || !Wo || !h || !ln2_w || !Wgu || !act || !ws || max_pages <= 0)
    return VLM_ERR_ARG;
  if (ldq % 8 != 0) return VLM_ERR_SHAPE;
  if (vlm_decode_block_supported(Hq, Hkv, D, inter, nsplit) != 1) return VLM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (bump_epoch) {
    hipLaunchKernelGGL(epoch_bump_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned*>((char*)ws + WS_EPOCH));
    VLM_CHECK_LAUNCH();
  }
  BlockArgs p{(const bf16_t*)q, (const bf16_t*)kpool, (const bf16_t*)vpool, (const int*)block_table, (const int*)kv_len,
              ldq, max_pages, Hkv, kv_len_add, nsplit, scale * 1.44269504088896340736f, (bf16_t*)part_o, (float*)part_ml,
              (const bf16_t*)Wo, (bf16_t*)h, (const bf16_t*)ln2_w, (const bf16_t*)Wgu, (bf16_t*)act, 2 * inter, eps, (char*)ws,
              (bump_epoch & 2) ? reinterpret_cast<u64*>((char*)ws + WS_STAMPS) : nullptr};
  const int G = Hq / Hkv;
#define GO2(GV, ID, LY, DP) hipLaunchKernelGGL((decode_block_kernel<GV, ID, KC_, LY, DP>), dim3(BLK_WGS), dim3(BLK_THREADS), 0, st, p)
#define GO1(GV, ID)                                                   \
  do {                                                                \
    if (g_layout == 0) { if (GV == 6) GO2(6, ID, 0, 64); else return VLM_ERR_SHAPE; } \
    else if (g_depth <= 4) GO2(GV, ID, 1, 4);                         \
    else if (g_depth <= 8) GO2(GV, ID, 1, 8);                         \
    else if (g_depth <= 16) GO2(GV, ID, 1, 16);                       \
    else GO2(GV, ID, 1, 64);                                          \
  } while (0)
#define GO(GV)                     \
  do {                             \
    if (!block_table) GO1(GV, true); \
    else GO1(GV, false);           \
  } while (0)
  switch (G) {
    case 3: GO(3); break;
    case 6: GO(6); break;
    default: return VLM_ERR_SHAPE;
  }
#undef GO
#undef GO1
#undef GO2
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

// -> 0 and the error word (hand-offs that gave up since the workspace was zeroed) / the 16 debug stamps (bump_epoch & 2)
extern "C" int vlm_decode_block_debug(const void* ws, unsigned* err, unsigned long long* stamps16) {
  if (!ws) return VLM_ERR_ARG;
  if (err && hipMemcpy(err, (const char*)ws + WS_ERR, 4, hipMemcpyDeviceToHost) != hipSuccess) return VLM_ERR_HIP;
  if (stamps16 && hipMemcpy(stamps16, (const char*)ws + WS_STAMPS, 16 * 8, hipMemcpyDeviceToHost) != hipSuccess) return VLM_ERR_HIP;
  return VLM_OK;
}
