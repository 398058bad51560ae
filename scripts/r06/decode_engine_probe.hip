// Round-6 probe (not product code): the COMPLETE persistent decode layer of MI355X_MICROARCH.md "engine-vs-launches" as a timing-faithful
// SKELETON at Qwen2-VL-2B dims - what would one launch for all 28 layers cost, before 1500 lines of real engine are written?
// ("build the guide's complete persistent layer once, timeboxed, or close the item ... done when frac >= 0.48 or the
// stamped timeline that shows it loses".)  Real memory behaviour, real dependency structure, dummy arithmetic:
//
//   256 workgroups (one per CU) x 4 waves, ONE launch for all 28 layers.
//   wave 0 = LOADER: streams this CU's slice of every weight matrix (qkv 8 rows, o_proj 6, gate/up 70, down 6 rows per layer:
//            357 KiB per layer and CU) into an LDS ring of 128 x 1 KiB pieces with non-temporal LDS-DMA, <= 32 pieces in flight,
//            runs ahead of the consumers by up to the ring (prefetch across every hop), publishes "landed" through LDS.
//   waves 1-3 = CONSUMERS: v_dot2-style dot products of their rows straight from the ring against the activation vector (registers for
//            K = 1536, LDS for K = 8960), wave reduction, results published as 8-byte {payload, tag} granules (sc1 stores).
//   hops (all by granule sweeps of ONE wave per workgroup, relaxed sc1 loads, every tag checked, bounded):
//     H1  h (768 granules)            -> every CU          (RMSNorm-like reduction, then the qkv rows)
//     H2  q / k / v (1024)            -> 16 attention units (kv head x 64-token page; the unit's K/V page was requested at layer start)
//     H3a partials (16 x 396)         -> the 16 units, each merges a 96-column slice of its kv head over the 8 pages
//     H3b merged attention (768)      -> every CU          (o_proj rows)
//     H4  h' (768)                    -> every CU          (gate/up rows: 210 of the layer's 357 pieces)
//     H5  act (4608)                  -> every CU          (down rows)
//   tags = (launch epoch << 8) | layer << 3 | hop; every (layer, hop) has its own granule region: nothing is re-armed inside a launch.
//
// Output: us per layer (whole launch / 28), the phase stamps of three workgroups, error count of the bounded waits (must be 0).
// Compare: the product's five launches per layer = 28.7 us per layer inside the 881.7 us step (profiles/r05_bench_kernel_stats.txt).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/r06/decode_engine_probe.hip -o scripts/r06/decode_engine_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int D = 1536, I = 8960, NQKV = 2048, NL = 28, NCU = 256;
constexpr int HKV = 2, PAGES = 8, UNITS = HKV * PAGES;          // ctx 512: 8 pages per kv head
constexpr int P_QKV = 24, P_O = 18, P_GU = 210, P_DN = 105, P_LAYER = P_QKV + P_O + P_GU + P_DN;   // 1 KiB pieces per layer and CU
constexpr int RING = 128;                                       // pieces in the LDS ring
constexpr int INFL = 32;                                        // pieces in flight
// granule regions of one layer (8-byte granules)
constexpr int G_H = 0, G_QKV = 768, G_PART = G_QKV + 1024, PART_STRIDE = 400, G_MRG = G_PART + UNITS * PART_STRIDE, G_H2 = G_MRG + 768,
              G_ACT = G_H2 + 768, G_LAYER = G_ACT + 4608;

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
__device__ __forceinline__ float dot8(u32x4_t w, u32x4_t x, float a) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w[i]), __builtin_bit_cast(bf16x2_t, x[i]), a, false);
  return a;
}
// wave-wide sum on the DPP data path (csrc/common.hpp: __shfl_xor is ds_bpermute, ~100 cycles per step - six dependent steps per
// reduction made the first version of this probe's consumers the bottleneck)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp<0xB1, 0xf>(0.f, v);
  v += dpp<0x4E, 0xf>(0.f, v);
  v += dpp<0x141, 0xf>(0.f, v);
  v += dpp<0x140, 0xf>(0.f, v);
  v += dpp<0x142, 0xa>(0.f, v);
  v += dpp<0x143, 0xc>(0.f, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// granules: ONE aligned 8-byte {payload, tag} agent-scope relaxed atomic store / load (lowers to global_store / global_load ... sc1);
// the compiler counts these loads itself (an inline-asm load's destination registers may be copied before the data lands when the
// kernel sits at its VGPR limit - the first version of this probe read garbage tags on some CUs that way)
__device__ __forceinline__ void gst(u64* p, unsigned payload, unsigned tag) {
  __hip_atomic_store(p, ((u64)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u32x2_t gld(const u64* p) {
  const u64 x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u32x2_t{(unsigned)x, (unsigned)(x >> 32)};
}

struct Params {
  const unsigned short* w;       // [NL] { Wqkv [2048][D] | Wo [D][D] | Wgu [2 I][D] | Wd [D][I] }
  const unsigned short* pages;   // [NL][UNITS][32 KB]
  u64* gran;                     // [NL + 1][G_LAYER]
  unsigned epoch;
  unsigned* err;
  u64* stamps;                   // [4 workgroups][NL][12]
  int thin;                      // 1: the loader keeps ONE fill in flight while its workgroup gathers
  int infl;                      // pieces in flight otherwise: 32 or 48
  int nohop;                     // 1: timing probe - no hop waits anything (every gather returns at once): the weight-stream ceiling of the engine
};

constexpr size_t LAYER_W = (size_t)NQKV * D + (size_t)D * D + (size_t)2 * I * D + (size_t)D * I;      // elements per layer

// LDS control words
struct Ctl {
  volatile int landed;          // pieces fully in LDS (stream index)
  volatile int progress[3];     // per consumer: first piece it still needs
  volatile int xready;          // (layer << 3 | op) + 1 of the newest activation vector in xbuf
  volatile int gathering;       // the gather wave is sweeping (loader thins)
};

// bounded LDS spin
__device__ __forceinline__ bool wait_ge(volatile int* p, int v, unsigned* err) {   // err: the counter of this wait site
  for (int it = 0; it < (1 << 22); ++it) {
    if (*p >= v) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  if ((threadIdx.x & 63) == 0) atomicAdd(err, 1u);
  return false;
}

// gather n granules starting at base (n multiple of 64 or padded region), every tag checked, -> xbuf[2 * idx .. ] as bf16 pairs
__device__ __forceinline__ void gather_to_lds(const u64* base, int n, unsigned tag, int lane, unsigned* xb, unsigned* err) {
  // cheap pre-poll on a spread sample, then full sweeps in batches of 16 loads per lane
  for (int it = 0; it < (1 << 18); ++it) {
    const u32x2_t v = gld(base + min(lane * (n >> 6), n - 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__all(v[1] == tag)) break;
    __builtin_amdgcn_s_sleep(4);
  }
  constexpr int GB = 24;      // granule loads per lane and sweep (12 KiB per sweep)
  for (int j0 = 0; j0 < n; j0 += 64 * GB) {
    for (int it = 0;; ++it) {
      u32x2_t v[GB];
#pragma unroll
      for (int j = 0; j < GB; ++j) v[j] = gld(base + min(j0 + lane + 64 * j, n - 1));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bool ok = true;
#pragma unroll
      for (int j = 0; j < GB; ++j) ok = ok && (v[j][1] == tag);
      if (__all(ok)) {
#pragma unroll
        for (int j = 0; j < GB; ++j)
          if (j0 + lane + 64 * j < n) xb[j0 + lane + 64 * j] = v[j][0];
        break;
      }
      if (it > (1 << 16)) { if (lane == 0) atomicAdd(err, 1u); break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
}

__global__ __launch_bounds__(256) void engine(Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                                   // 128 KiB
  unsigned* xb = reinterpret_cast<unsigned*>(smem + RING * 1024);      // activation vector as bf16 pairs: up to 4608 words (18 KiB)
  Ctl* ctl = reinterpret_cast<Ctl*>(smem + RING * 1024 + 4608 * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), cu = blockIdx.x;
  if (tid == 0) { ctl->landed = 0; ctl->progress[0] = ctl->progress[1] = ctl->progress[2] = 0; ctl->xready = 0; ctl->gathering = 0; }
  __syncthreads();
  const int sidx = cu == 0 ? 0 : cu == 5 ? 1 : cu == 100 ? 2 : cu == 255 ? 3 : -1;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  auto stamp = [&](int layer, int k) {
    const bool unit_stamp = k >= 2 && k <= 4;
    if (sidx >= 0 && lane == 0 && (unit_stamp ? wave == 3 : wave == 1)) p.stamps[((size_t)sidx * NL + layer) * 12 + k] = __builtin_amdgcn_s_memrealtime() - t0;
  };

  if (wave == 0) {
    // ------------------------------------------------------------------------------------------------ loader
    // Per segment, in FILLS of <= 16 pieces (16 KiB, the guide's ring slot): one ring-space check, the LDS-DMA instructions back to
    // back, one counted wait per fill: at most INFL pieces in flight (16 while the workgroup gathers, when thinning is on).
    int gp = 0;      // stream piece index
    int pr_seen = 0;
    for (int layer = 0; layer < NL; ++layer) {
      const unsigned short* wl = p.w + (size_t)layer * LAYER_W;
      const char* seg[4] = {reinterpret_cast<const char*>(wl + (size_t)cu * 8 * D), reinterpret_cast<const char*>(wl + (size_t)NQKV * D + (size_t)cu * 6 * D),
                            reinterpret_cast<const char*>(wl + (size_t)NQKV * D + (size_t)D * D + (size_t)cu * 70 * D),
                            reinterpret_cast<const char*>(wl + (size_t)NQKV * D + (size_t)D * D + (size_t)2 * I * D + (size_t)cu * 6 * I)};
      const int np[4] = {P_QKV, P_O, P_GU, P_DN};
      for (int sg = 0; sg < 4; ++sg) {
        for (int i0 = 0; i0 < np[sg]; i0 += 16) {
          const int nf = min(16, np[sg] - i0);
          if (gp + nf - RING > pr_seen) {          // (the consumers' progress is re-read only when the cached value does not clear the fill)
            for (int it = 0;; ++it) {
              pr_seen = min(ctl->progress[0], min(ctl->progress[1], ctl->progress[2]));
              if (gp + nf - RING <= pr_seen) break;
              if (it == 0) {     // ring full: everything issued has to be visible, or a consumer waiting for a piece in flight never moves
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) ctl->landed = gp;
              }
              __builtin_amdgcn_s_sleep(2);
              if (it > (1 << 22)) { if (lane == 0) atomicAdd(p.err + 6, 1u); break; }
            }
          }
          const char* src = seg[sg] + (size_t)i0 * 1024 + lane * 16;
          const unsigned slot0 = (unsigned)gp & (RING - 1);
          if (nf == 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 1024),
                                               (__attribute__((address_space(3))) void*)(ring + ((slot0 + k) & (RING - 1)) * 1024), 16, 0, 2 /* nt */);
          } else {
            for (int k = 0; k < nf; ++k)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + k * 1024),
                                               (__attribute__((address_space(3))) void*)(ring + ((slot0 + k) & (RING - 1)) * 1024), 16, 0, 2 /* nt */);
          }
          gp += nf;
          if (p.thin && ctl->gathering) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (lane == 0) ctl->landed = gp - 16;
          } else if (p.infl == 48) {
            asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
            if (lane == 0) ctl->landed = gp - 48;
          } else {
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            if (lane == 0) ctl->landed = gp - 32;
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) ctl->landed = gp;
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  const int cw = wave - 1;                         // 0..2
  if (p.nohop == 2) {                              // loader-only probe: the ring is always free, nobody reads it
    if (lane == 0) ctl->progress[cw] = 0x7fffffff;
    return;
  }
  const bool is_unit = cu < UNITS && cw == 2;      // wave 3 of CUs 0..15: an attention unit (kv head cu / 8, page cu % 8)
  int piece0 = 0;                                  // stream piece where the current op's segment starts
  for (int layer = 0; layer < NL; ++layer) {
    u64* gl = p.gran + (size_t)layer * G_LAYER;
    u64* gn = p.gran + (size_t)(layer + 1) * G_LAYER;
    const unsigned tagb = (p.epoch << 8) | (unsigned)(layer << 3);
    // the attention unit requests its K/V page now (old tokens do not depend on this step)
    u32x4_t pg[32];
    if (is_unit) {
      const unsigned short* pgp = p.pages + ((size_t)layer * UNITS + cu) * 16384;
#pragma unroll
      for (int i = 0; i < 32; ++i) pg[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(pgp + (size_t)(i * 64 + lane) * 8));
    }

    // one GEMV op over `rows` rows of K elements starting at stream piece `pc0`; rows in units of `ru`, unit u -> wave u % 3;
    // x from registers (K = 1536) or LDS (K = 8960); publishes one granule per unit at gout[cu * nunits + u]
    auto gemv_op = [&](int pc0, int rows, int K, int ru, u64* gout, unsigned tag, int xr_token) {
      wait_ge(&ctl->xready, xr_token, p.err + 1);
      u32x4_t xr[3];
      float ss = 0.f;
      if (K == D) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          xr[c] = *reinterpret_cast<const u32x4_t*>(xb + (lane + 64 * c) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) ss += bf_lo(xr[c][i]) * bf_lo(xr[c][i]) + bf_hi(xr[c][i]) * bf_hi(xr[c][i]);
        }
        ss = rsqrtf(wave_sum(ss) / (float)D + 1e-6f);      // RMSNorm-like statistic (used as a scale of the result)
      } else {
        ss = 1.f;
      }
      const int nunits = (rows + ru - 1) / ru, cpr = K / 8;       // 16-byte chunks per row
      for (int u = cw; u < nunits; u += 3) {
        const long b0 = (long)pc0 * 1024 + (long)u * ru * K * 2;   // stream byte of the unit's first row
        const int nr = min(ru, rows - u * ru);
        if (lane == 0) ctl->progress[cw] = (int)(b0 >> 10);
        const long b1 = b0 + (long)nr * K * 2;
        wait_ge(&ctl->landed, (int)((b1 + 1023) >> 10), p.err + 0);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < nr; ++r) {
          const long rb = b0 + (long)r * K * 2;
          if (K == D) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const u32x4_t w = *reinterpret_cast<const u32x4_t*>(ring + ((rb + (lane + 64 * c) * 16) & (RING * 1024 - 1)));
              acc[r] = dot8(w, xr[c], acc[r]);
            }
          } else {
            for (int c = lane; c < cpr; c += 64) {
              const u32x4_t w = *reinterpret_cast<const u32x4_t*>(ring + ((rb + c * 16) & (RING * 1024 - 1)));
              const u32x4_t x = *reinterpret_cast<const u32x4_t*>(xb + c * 4);
              acc[r] = dot8(w, x, acc[r]);
            }
          }
        }
        float v0 = wave_sum(acc[0]) * ss, v1 = wave_sum(acc[1]) * ss, v2 = wave_sum(acc[2]), v3 = wave_sum(acc[3]);
        if (ru == 4) { v0 = v0 / (1.f + __expf(-v0)) * v1; v1 = v2 / (1.f + __expf(-v2)) * v3; }      // SwiGLU-like
        if (lane == 0) gst(gout + cu * nunits + u, pack2(v0, v1), tag);
      }
      if (lane == 0) ctl->progress[cw] = pc0 + (rows * K * 2 + 1023) / 1024;
    };
    // gather by wave 1 (cw == 0): sweep -> xbuf -> xready token
    auto gather_all = [&](const u64* src, int n, unsigned tag, int token, int prev_end) {
      if (p.nohop) {
        if (cw == 0 && lane == 0) ctl->xready = token;
        return;
      }
      if (cw == 0) {
        // xbuf is overwritten: every consumer must be done with the previous op (its progress at the op's end)
        wait_ge(&ctl->progress[1], prev_end, p.err + 2);
        wait_ge(&ctl->progress[2], prev_end, p.err + 2);
        if (lane == 0) ctl->gathering = 1;
        gather_to_lds(src, n, tag, lane, xb, p.err + 3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) { ctl->gathering = 0; ctl->xready = token; }
      }
    };
    const int tk = layer * 8;

    // ---- H1: h -> every CU; qkv rows
    gather_all(gl + G_H, 768, tagb | 0, tk + 1, piece0);
    stamp(layer, 0);
    gemv_op(piece0, 8, D, 2, gl + G_QKV, tagb | 1, tk + 1);
    stamp(layer, 1);
    // ---- attention units: H2 (q of the kv head + k / v of the new token), 32 MFMAs, partial; H3a slice merge
    if (is_unit && !p.nohop) {
      const int g = cu / PAGES, s = cu % PAGES;
      unsigned qg[8];
      for (int it = 0;; ++it) {
        u32x2_t v[8];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = gld(gl + G_QKV + g * 384 + lane + 64 * j);
#pragma unroll
        for (int j = 0; j < 2; ++j) v[6 + j] = gld(gl + G_QKV + 768 + g * 128 + lane + 64 * j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ok = ok && v[j][1] == (tagb | 1); qg[j] = v[j][0]; }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if (it > (1 << 18)) { if (lane == 0) atomicAdd(p.err + 4, 1u); break; }
      }
      stamp(layer, 2);
      bf16x8_t qf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4_t t = {qg[i], qg[(i + 1) % 6], qg[(i + 2) % 6] ^ qg[6], qg[(i + 3) % 6] ^ qg[7]};
        qf[i] = __builtin_bit_cast(bf16x8_t, t);
      }
      f32x4_t st[4], acc[8];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        st[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pg[t * 4 + ds]), qf[ds], st[t], 0, 0, 0);
      }
      float mx = -1e30f, ls = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[t][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[t][r] = exp2f((st[t][r] - mx) * 1e-3f); ls += st[t][r]; }
      bf16x8_t pb[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const u32x4_t t = {pack2(st[2 * u][0], st[2 * u][1]), pack2(st[2 * u][2], st[2 * u][3]), pack2(st[2 * u + 1][0], st[2 * u + 1][1]),
                           pack2(st[2 * u + 1][2], st[2 * u + 1][3])};
        pb[u] = __builtin_bit_cast(bf16x8_t, t);
      }
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        acc[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pg[16 + dt * 2 + u]), pb[u], acc[dt], 0, 0, 0);
      }
      const int head = lane & 15, gq = lane >> 4;
      u64* gpp = gl + G_PART + (size_t)cu * PART_STRIDE;
      if (head < 6) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          gst(gpp + head * 64 + dt * 8 + gq * 2, pack2(acc[dt][0], acc[dt][1]), tagb | 2);
          gst(gpp + head * 64 + dt * 8 + gq * 2 + 1, pack2(acc[dt][2], acc[dt][3]), tagb | 2);
        }
        if (gq == 0) {
          gst(gpp + 384 + head * 2, __float_as_uint(mx), tagb | 2);
          gst(gpp + 384 + head * 2 + 1, __float_as_uint(ls), tagb | 2);
        }
      }
      stamp(layer, 3);
      // H3a: my 48-granule column slice of the 8 pages of my kv head + their (m, l): 8 x 48 + 8 x 12 = 480 items, 8 per lane
      unsigned part[8];
      for (int it = 0;; ++it) {
        u32x2_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int item = lane + 64 * k;            // < 512
          const int sj = min(item / 60, PAGES - 1), c = item % 60;
          v[k] = gld(gl + G_PART + (size_t)(g * PAGES + sj) * PART_STRIDE + (c < 48 ? 48 * s + c : 384 + (c - 48)));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) { ok = ok && v[k][1] == (tagb | 2); part[k] = v[k][0]; }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if (it > (1 << 18)) { if (lane == 0) atomicAdd(p.err + 5, 1u); break; }
      }
      float mv = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) mv += bf_lo(part[k]) * 0.125f + bf_hi(part[k]);
      mv += __shfl_xor(mv, 32, 64);
      if (lane < 48) gst(gl + G_MRG + g * 384 + 48 * s + lane, pack2(mv, mv * 0.5f), tagb | 3);
      stamp(layer, 4);
    }
    // ---- H3b: merged attention -> every CU; o_proj rows
    gather_all(gl + G_MRG, 768, tagb | 3, tk + 2, piece0 + P_QKV);
    stamp(layer, 5);
    gemv_op(piece0 + P_QKV, 6, D, 2, gl + G_H2, tagb | 4, tk + 2);
    stamp(layer, 6);
    // ---- H4: h' -> every CU; gate/up rows
    gather_all(gl + G_H2, 768, tagb | 4, tk + 3, piece0 + P_QKV + P_O);
    stamp(layer, 7);
    gemv_op(piece0 + P_QKV + P_O, 70, D, 4, gl + G_ACT, tagb | 5, tk + 3);
    stamp(layer, 8);
    // ---- H5: act -> every CU; down rows -> the NEXT layer's h
    gather_all(gl + G_ACT, 4608, tagb | 5, tk + 4, piece0 + P_QKV + P_O + P_GU);
    stamp(layer, 9);
    gemv_op(piece0 + P_QKV + P_O + P_GU, 6, I, 2, gn + G_H, ((p.epoch << 8) | (unsigned)((layer + 1) << 3)) | 0, tk + 4);
    stamp(layer, 10);
    piece0 += P_LAYER;
  }
}

// the first layer's h: 768 granules tagged for layer 0
__global__ void seed_h(u64* gran, unsigned epoch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 768) gst(gran + G_H + i, 0x3c003c00u, (epoch << 8) | 0u);
}

int main(int argc, char** argv) {
  const int thin = argc > 1 ? atoi(argv[1]) : 1, infl = argc > 2 ? atoi(argv[2]) : 48, nohop = argc > 3 ? atoi(argv[3]) : 0;
  const size_t wbytes = (size_t)NL * LAYER_W * 2, pbytes = (size_t)NL * UNITS * 32768, gbytes = (size_t)(NL + 1) * G_LAYER * 8;
  unsigned short *w, *pages;
  u64 *gran, *stamps;
  unsigned* err;
  CK(hipMalloc(&w, wbytes)); CK(hipMalloc(&pages, pbytes)); CK(hipMalloc(&gran, gbytes)); CK(hipMalloc(&err, 32));
  CK(hipMalloc(&stamps, 4 * NL * 12 * 8));
  CK(hipMemset(w, 0x3c, wbytes)); CK(hipMemset(pages, 0x3c, pbytes)); CK(hipMemset(gran, 0, gbytes)); CK(hipMemset(err, 0, 32));
  CK(hipMemset(stamps, 0, 4 * NL * 12 * 8));
  const int lds = RING * 1024 + 4608 * 4 + 64;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&engine), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, engine, 256, lds));
  printf("weights %.2f GB, %d pieces per layer and CU, LDS %d B, occupancy %d workgroup(s) per CU, loader thinning %d, %d pieces in flight\n", wbytes / 1e9, P_LAYER, lds, nb, thin, infl);
  if (nb < 1) return 1;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned epoch = 1;
  float best = 1e30f;
  for (int rep = 0; rep < 8; ++rep, ++epoch) {
    Params p{w, pages, gran, epoch, err, stamps, thin, infl, nohop};
    hipLaunchKernelGGL(seed_h, dim3(3), dim3(256), 0, st, gran, epoch);
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(engine, dim3(NCU), dim3(256), lds, st, p);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned he[8];
    CK(hipMemcpy(he, err, 32, hipMemcpyDeviceToHost));
    unsigned herr = 0;
    for (int i = 0; i < 7; ++i) herr += he[i];
    printf("launch %d: %8.1f us = %6.2f us per layer (bounded-wait failures: landed %u xready %u guard %u sweep %u unit-q %u unit-slice %u ring %u)\n", rep,
           ms * 1e3, ms * 1e3 / NL, he[0], he[1], he[2], he[3], he[4], he[5], he[6]);
    if (rep >= 2 && ms < best) best = ms;
    if (herr) break;
  }
  printf("best: %.2f us per layer; 28 layers + 82 us of head / tail = %.1f us per token = %.0f tok/s (the five-launch step: 881.7 us, 1134 tok/s)\n",
         best * 1e3 / NL, best * 1e3 + 82.0, 1e6 / (best * 1e3 + 82.0));
  std::vector<u64> hs(4 * NL * 12);
  CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
  const char* names[11] = {"H1 h gathered", "qkv published", "unit: q gathered", "unit: partial published", "unit: slice published",
                           "H3b attention gathered", "o_proj published", "H4 h' gathered", "gate/up published", "H5 act gathered", "down published"};
  const int cus[4] = {0, 5, 100, 255};
  for (int layer : {0, 1, 14}) {
    printf("layer %d, us since the layer's first stamp on that workgroup (wave 1 of CU 0 / 5 / 100 / 255; unit stamps: wave 3 of CU 0 / 5):\n", layer);
    for (int k = 0; k < 11; ++k) {
      printf("  %-26s", names[k]);
      for (int s = 0; s < 4; ++s) {
        const u64 v = hs[((size_t)s * NL + layer) * 12 + k], b = hs[((size_t)s * NL + layer) * 12 + 0];
        if (v) printf(" %8.2f", (double)((long long)v - (long long)b) * 0.01); else printf("        -");
      }
      printf("\n");
    }
  }
  (void)cus;
  return 0;
}
