// One launch for the back half of a decoder layer at batch 1 on gfx950:
//     h   += attn . Wo^T                                (o_proj + residual,            language.py:115-120,151)
//     act  = swiglu(RMSNorm(h) . Wgate/up^T)             (post-attention norm + MLP in, language.py:130-133,152; mlp.py:6-14)
//     h   += act . Wdown^T                               (MLP out + residual,           language.py:153)
// i.e. three of the five dependent kernels of the decode layer (csrc/engine.hip) with their two all-to-all
// hand-offs moved INSIDE the launch.
//
// Why (measured on MI355X, profiles/r02_*): the three GEMV kernels cost 3.4 + 10.6 + 7.0 us of the step's wall
// clock per layer for 87 MB of weights (12.8 us of HBM time): each pays a kernel boundary (1.5 us) and a cold first
// touch before its stream starts, and its stream cannot start before its input vector exists.  Here every
// workgroup issues its whole o_proj and gate/up weight slice into REGISTERS at entry (the register file of the
// chip is 128 MB; a layer's weights are 94 MB), so the HBM streams from the first cycle while the hand-offs
// happen; the down slice is issued as soon as the gate/up registers are free.
//
// Structure: 256 workgroups (one per CU, all resident: 8 waves x <= 256 VGPRs) x 512 threads.
//   * waves 0-6 are workers: they own rows of Wo (1 row per wave), gate/up (row pairs) and a K range of Wdown and
//     hold those weights in VGPRs.  Their accumulation order is that of the unfused kernels (row-wave GEMV), so
//     h after o_proj and act are bit-identical to the unfused path; the down projection splits K over 7 waves
//     instead of 4 (fp32 summation order differs).
//   * wave 7 is the gatherer.  A worker wave that has 30 weight loads in flight cannot poll anything - vector
//     loads return in issue order, the poll would come back behind the whole weight stream - so the wave that
//     sweeps the hand-off buffers carries no weights.  It gathers the vector every workgroup needs (h after
//     o_proj: 1536 values, act: 8960 values) into LDS and releases the workers through a workgroup barrier.
//   * hand-off = data-tagged granules (cdna_hip_programming.md Guideline 16, form R2): each value is ONE
//     naturally aligned 4-byte agent-scope store {bf16 value, 16-bit tag}; the tag is the launch epoch of this
//     layer's buffers, so a granule is valid iff its tag matches - no flag, no fence, no counter, nothing to
//     reset.  Consumers read with 16-byte sc1 loads (bypass the per-CU L1) and re-read until every tag matches.
//     Results never depend on placement or timing; every spin is bounded (err word set, garbage out, no hang).
#include "common.cuh"
#include "internal.h"
#include "../../include/vlm_hip.h"

namespace {

constexpr int FM_NWG = 256, FM_THREADS = 512, FM_WORKERS = 7, FM_WT = FM_WORKERS * 64;

__device__ __forceinline__ u32x4_t ntl16(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
}
__device__ __forceinline__ float fdot2(unsigned w, unsigned x, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, x), acc, false);
}
// same order as gemv_bf16.hip::dot8 (bit-identical partial sums)
__device__ __forceinline__ float fdot8(const u32x4_t w, const u32x4_t x, float acc) {
  const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
  acc = fdot2(w0, x0, acc);
  acc = fdot2(w1, x1, acc);
  acc = fdot2(w2, x2, acc);
  acc = fdot2(w3, x3, acc);
  return acc;
}

// 8 granules (32 bytes) -> 8 bf16 packed in 16 bytes; ok = every tag matches
__device__ __forceinline__ u32x4_t unpack8(const u32x4_t a, const u32x4_t b, unsigned tag, bool& ok) {
  ok = ok && (a[0] >> 16) == tag && (a[1] >> 16) == tag && (a[2] >> 16) == tag && (a[3] >> 16) == tag &&
       (b[0] >> 16) == tag && (b[1] >> 16) == tag && (b[2] >> 16) == tag && (b[3] >> 16) == tag;
  u32x4_t o;
  o[0] = (a[0] & 0xffffu) | (a[1] << 16);
  o[1] = (a[2] & 0xffffu) | (a[3] << 16);
  o[2] = (b[0] & 0xffffu) | (b[1] << 16);
  o[3] = (b[2] & 0xffffu) | (b[3] << 16);
  return o;
}

struct FusedMlpArgs {
  const bf16_t* attn;     // [KO] attention output (o_proj input)
  bf16_t* h;              // [D] residual stream, in place
  const bf16_t *wo, *ln2_w, *wgu, *wdown;
  unsigned* g_h;          // [D] granules: h after o_proj
  unsigned* g_act;        // [I] granules: swiglu output
  unsigned* epoch;        // [1] launch counter of these buffers (starts at 1; the launch increments it)
  unsigned* err;          // [1] set when a bounded spin gave up
  float eps;
  int D, I, KO;
  float* stamps;          // debug timeline of workgroup 0 (16 floats, microseconds since its start) or nullptr
  int mode;               // measurement knobs: bit 0 = do not wait for the hand-offs (timing only, results garbage);
                          // bit 1 = the gatherer starts sweeping only after its own workgroup has published; bits 8..15 =
                          // s_sleep units between sweep passes (0 = 2)
};

// KC = D / 512, KCO = KO / 512 (16-byte chunks per lane of one row), MAXP = ceil((I / 256) / 7) row pairs per worker,
// DNL = ceil((I / 8) / 448) down chunks per worker thread and row, RD = D / 256 rows of Wo / Wdown per workgroup
template <int KC, int KCO, int MAXP, int DNL, int RD>
__global__ __launch_bounds__(FM_THREADS, 2) void mlp_fused_kernel(FusedMlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: xs [D] bf16 normalised h | hm [D] bf16 raw h after o_proj | act [I] bf16 | red [7][RD] f32
  u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem);
  u32x4_t* hm = reinterpret_cast<u32x4_t*>(smem + (size_t)a.D * 2);
  u32x4_t* acts = reinterpret_cast<u32x4_t*>(smem + (size_t)a.D * 4);
  float* red = reinterpret_cast<float*>(smem + (size_t)a.D * 4 + (size_t)a.I * 2);
  volatile int* lflag = reinterpret_cast<volatile int*>(red + FM_WORKERS * RD);     // [2] own-workgroup progress (LDS)
  const int nap = ((a.mode >> 8) & 0xff) ? ((a.mode >> 8) & 0xff) : 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
  const unsigned tag = (*a.epoch & 0x7fffu) | 0x8000u;        // never 0 (zero-initialised buffers), differs from launch to launch
  const int P = a.I / FM_NWG;                                   // gate/up row pairs (= act outputs) per workgroup
  const int nchD = a.D >> 3, nchO = a.KO >> 3, nchI = a.I >> 3;
  if (tid == 0) { lflag[0] = 0; lflag[1] = 0; }
  __syncthreads();
  const unsigned long long t_start = a.stamps ? wall_clock64() : 0ull;
  auto stamp = [&](int slot) {     // wave 0 -> slots 0..7, gatherer -> 8..15 (100 MHz clock)
    if (a.stamps && wg == 0 && lane == 0 && (wave == 0 || wave == FM_WORKERS))
      a.stamps[slot + (wave == 0 ? 0 : 8)] = (float)(wall_clock64() - t_start) * 0.01f;
  };

  if (wave < FM_WORKERS) {
    // =================================================================== workers
    // ---- small loads first (they return first): o_proj input chunks and the residual element
    u32x4_t xo[KCO];
#pragma unroll
    for (int c = 0; c < KCO; ++c) xo[c] = *reinterpret_cast<const u32x4_t*>(a.attn + (size_t)min(lane + 64 * c, nchO - 1) * 8);
    const int row_o = wg * RD + min(wave, RD - 1);
    const bf16_t res_o = a.h[row_o];
    __builtin_amdgcn_sched_barrier(0);
    // ---- the weight stream: Wo row, then the gate/up rows - all in flight before any arithmetic
    u32x4_t wo_r[KCO];
#pragma unroll
    for (int c = 0; c < KCO; ++c) wo_r[c] = ntl16(a.wo + (size_t)row_o * a.KO + (size_t)min(lane + 64 * c, nchO - 1) * 8);
    const int base_p = P / FM_WORKERS, extra = P % FM_WORKERS;
    const int my_cnt = base_p + (wave < extra ? 1 : 0);
    const int my_p0 = wg * P + wave * base_p + min(wave, extra);                 // first pair (= act index) of this wave
    u32x4_t wgu_r[MAXP][2][KC];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int pair = my_p0 + min(p, my_cnt - 1);                                 // clamped: surplus slots re-read the last pair
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < KC; ++c)
          wgu_r[p][r][c] = ntl16(a.wgu + ((size_t)2 * pair + r) * a.D + (size_t)min(lane + 64 * c, nchD - 1) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- o_proj row + residual -> granule
    {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < KCO; ++c) {
        const u32x4_t xv = (lane + 64 * c < nchO) ? xo[c] : u32x4_t{0, 0, 0, 0};
        acc = fdot8(wo_r[c], xv, acc);
      }
      acc = wave_sum(acc);
      if (wave < RD && lane == 0) {
        const float v = rbf(acc) + bf2f(res_o);
        __hip_atomic_store(a.g_h + row_o, (tag << 16) | (unsigned)f2bf(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    stamp(0);                                           // o_proj row published
    if (wave == 0 && lane == 0) lflag[0] = 1;
    __syncthreads();                                    // #1: xs / hm are in LDS (gatherer)
    stamp(1);
    // ---- gate/up rows on the normalised vector, SwiGLU -> granules
    {
      u32x4_t xn[KC];
#pragma unroll
      for (int c = 0; c < KC; ++c) xn[c] = (lane + 64 * c < nchD) ? xs[lane + 64 * c] : u32x4_t{0, 0, 0, 0};
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        float g = 0.f, u = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          g = fdot8(wgu_r[p][0][c], xn[c], g);
          u = fdot8(wgu_r[p][1][c], xn[c], u);
        }
        g = wave_sum(g);
        u = wave_sum(u);
        if (p < my_cnt && lane == 0) {
          const unsigned o = (unsigned)f2bf(swiglu_(rbf(g), rbf(u)));
          __hip_atomic_store(a.g_act + my_p0 + p, (tag << 16) | o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    stamp(2);                                           // gate/up rows published
    if (wave == 0 && lane == 0) lflag[1] = 1;
    // ---- the down slice goes out now (the gate/up registers are free): rows wg*RD .. +RD-1, K split over the 448
    //      worker threads
    u32x4_t wd_r[RD][DNL];
#pragma unroll
    for (int r = 0; r < RD; ++r)
#pragma unroll
      for (int i = 0; i < DNL; ++i)
        wd_r[r][i] = ntl16(a.wdown + (size_t)(wg * RD + r) * a.I + (size_t)min(tid + FM_WT * i, nchI - 1) * 8);
    __syncthreads();                                    // #2: act is in LDS (gatherer)
    stamp(3);
    {
      float acc[RD];
#pragma unroll
      for (int r = 0; r < RD; ++r) acc[r] = 0.f;
#pragma unroll
      for (int i = 0; i < DNL; ++i) {
        const int ch = tid + FM_WT * i;
        const u32x4_t xv = ch < nchI ? acts[min(ch, nchI - 1)] : u32x4_t{0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < RD; ++r) acc[r] = fdot8(wd_r[r][i], xv, acc[r]);
      }
#pragma unroll
      for (int r = 0; r < RD; ++r) {
        const float s = wave_sum(acc[r]);
        if (lane == 0) red[wave * RD + r] = s;
      }
    }
    stamp(4);                                           // down partial sums done (weights had landed)
    __syncthreads();                                    // #3: partial sums of the 7 worker waves
    if (tid < RD) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < FM_WORKERS; ++w) v += red[w * RD + tid];
      const int row = wg * RD + tid;
      const bf16_t hres = reinterpret_cast<const bf16_t*>(hm)[row];
      a.h[row] = f2bf(rbf(v) + bf2f(hres));
    }
    stamp(5);
  } else {
    // =================================================================== gatherer (wave 7): no weight loads
    const auto rs_h = __builtin_amdgcn_make_buffer_rsrc(a.g_h, 0, a.D * 4, 0x00020000);
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(a.g_act, 0, a.I * 4, 0x00020000);
    uint4 nwv[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) nwv[c] = reinterpret_cast<const uint4*>(a.ln2_w)[min(lane + 64 * c, nchD - 1)];
    // ---- hand-off 1: h after o_proj, every row from its producer workgroup
    u32x4_t hv[KC];
    if (a.mode & 2) { while (lflag[0] == 0) __builtin_amdgcn_s_sleep(4); }
    {
      unsigned spins = 0;
      for (;;) {
        u32x4_t ga[KC], gb[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const int off = min(lane + 64 * c, nchD - 1) * 32;
          ga[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_h, off, 0, 16);       // aux 16 = sc1
          gb[c] = __builtin_amdgcn_raw_buffer_load_b128(rs_h, off + 16, 0, 16);
        }
        bool ok = true;
#pragma unroll
        for (int c = 0; c < KC; ++c) hv[c] = unpack8(ga[c], gb[c], tag, ok);
        if (__all(ok) || (a.mode & 1)) break;
        if (++spins > (1u << 20)) { if (lane == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(1);
      }
    }
    stamp(0);                                           // hand-off 1 complete (h after o_proj from all workgroups)
    // RMSNorm exactly as the prologue of gemv_rowwave_kernel (same lane <-> chunk map, same order of operations)
    {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        if (lane + 64 * c >= nchD) hv[c] = u32x4_t{0, 0, 0, 0};
        const float v[8] = {bf_lo(hv[c][0]), bf_hi(hv[c][0]), bf_lo(hv[c][1]), bf_hi(hv[c][1]),
                            bf_lo(hv[c][2]), bf_hi(hv[c][2]), bf_lo(hv[c][3]), bf_hi(hv[c][3])};
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j] * v[j];
      }
      const float inv = rsqrtf(wave_sum(s) / (float)a.D + a.eps);
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int ch = lane + 64 * c;
        const u32x4_t u = hv[c];
        const uint4 wu = nwv[c];
        u32x4_t o;
        o[0] = pack_bf2(bf_lo(wu.x) * rbf(bf_lo(u[0]) * inv), bf_hi(wu.x) * rbf(bf_hi(u[0]) * inv));
        o[1] = pack_bf2(bf_lo(wu.y) * rbf(bf_lo(u[1]) * inv), bf_hi(wu.y) * rbf(bf_hi(u[1]) * inv));
        o[2] = pack_bf2(bf_lo(wu.z) * rbf(bf_lo(u[2]) * inv), bf_hi(wu.z) * rbf(bf_hi(u[2]) * inv));
        o[3] = pack_bf2(bf_lo(wu.w) * rbf(bf_lo(u[3]) * inv), bf_hi(wu.w) * rbf(bf_hi(u[3]) * inv));
        if (ch < nchD) { xs[ch] = o; hm[ch] = u; }
      }
    }
    __syncthreads();                                    // #1
    stamp(1);
    // ---- hand-off 2: act.  While the producers are still streaming their gate/up rows only one granule per producer
    //      workgroup is polled (1 KB per pass instead of 35 KB); the full sweep starts once all of them have shown up.
    if (a.mode & 2) { while (lflag[1] == 0) __builtin_amdgcn_s_sleep(8); }
    {
      unsigned spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < FM_NWG / 64; ++k) {
          const unsigned g = __hip_atomic_load(a.g_act + (size_t)(lane * (FM_NWG / 64) + k) * P, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && (g >> 16) == tag;
        }
        if (__all(ok) || (a.mode & 1)) break;
        if (++spins > (1u << 20)) { if (lane == 0) __hip_atomic_store(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(4);
      }
      stamp(2);                                         // every producer's first act granule has arrived
      const int nit = (nchI + 63) / 64;
      int done = 0;                                      // iterations [0, done) are complete and already in LDS
      spins = 0;
      while (done < nit) {
        // sweep up to 8 iterations (16 loads) per pass, starting at the first incomplete one
        u32x4_t ga[8], gb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int ch = min(lane + 64 * min(done + q, nit - 1), nchI - 1);
          ga[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ch * 32, 0, 16);
          gb[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ch * 32 + 16, 0, 16);
        }
        int adv = 0;
        bool still = true;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          bool ok = true;
          const u32x4_t o = unpack8(ga[q], gb[q], tag, ok);
          const int it = done + q, ch = lane + 64 * it;
          const bool valid = it < nit && ch < nchI;
          const bool all_ok = __all(ok || !valid) != 0 || (a.mode & 1);
          if (still && it < nit && all_ok) {
            if (valid) acts[ch] = o;
            ++adv;
          } else {
            still = false;
          }
        }
        done += adv;
        if (adv == 0) {
          if (++spins > (1u << 20)) { if (lane == 0) __hip_atomic_store(a.err, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    stamp(3);                                           // hand-off 2 complete (act in LDS)
    __syncthreads();                                    // #2
    __syncthreads();                                    // #3
    if (wg == 0 && lane == 0) *a.epoch = *a.epoch + 1;  // next launch of this layer: a new tag (visible at the kernel boundary)
  }
}

}  // namespace

int vlm_mlp_fused_supported(int D, int I, int KO) {
  // register-resident weight slices: the instantiations below (Qwen2-VL-2B: 1536 / 8960 / 1536; the 0.5B decoder of
  // nanoLLaVA with 128-wide padded heads: 1024 / 2816 / 2048)
  return (D == 1536 && I == 8960 && KO == 1536) || (D == 1024 && I == 2816 && KO == 2048);
}

int vlm_mlp_fused_launch(const void* attn, void* h, const void* wo, const void* ln2_w, const void* wgu, const void* wdown,
                         void* g_h, void* g_act, void* epoch, void* err, float eps, int D, int I, int KO, void* stamps,
                         int mode, void* stream) {
  if (!attn || !h || !wo || !ln2_w || !wgu || !wdown || !g_h || !g_act || !epoch || !err) return VLM_ERR_ARG;
  if (!vlm_mlp_fused_supported(D, I, KO)) return VLM_ERR_SHAPE;
  FusedMlpArgs a{(const bf16_t*)attn, (bf16_t*)h, (const bf16_t*)wo, (const bf16_t*)ln2_w, (const bf16_t*)wgu,
                 (const bf16_t*)wdown, (unsigned*)g_h, (unsigned*)g_act, (unsigned*)epoch, (unsigned*)err, eps, D, I, KO,
                 (float*)stamps, mode};
  const int RD = D / FM_NWG;
  const size_t lds = (size_t)D * 4 + (size_t)I * 2 + (size_t)FM_WORKERS * RD * sizeof(float) + 16;
  hipStream_t st = (hipStream_t)stream;
  if (D == 1536)
    hipLaunchKernelGGL((mlp_fused_kernel<3, 3, 5, 3, 6>), dim3(FM_NWG), dim3(FM_THREADS), lds, st, a);
  else
    hipLaunchKernelGGL((mlp_fused_kernel<2, 4, 2, 1, 4>), dim3(FM_NWG), dim3(FM_THREADS), lds, st, a);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
