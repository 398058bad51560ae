// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// Replaces every nn.Linear on the prefill path of the reference
// (mlx_vlm/models/qwen2_vl/vision.py:129-130,137,161,168-173,110-120;
//  language.py:52-55,76,120; mlp.py:9-14; language.py:514-517 as_linear) and the
// stride==kernel Conv3d patch projection (vision.py:83-101), which is a GEMM.
//
// Design (CDNA4): both operands are K-contiguous ([rows][K]), so A and W
// fragments are the same "row x 8 consecutive k" 16-byte reads.  128x128 (or
// 64-wide) block tile, BK = 64, 4 waves (2x2), each wave 4x4 (or fewer) tiles of
// v_mfma_f32_16x16x32_bf16 accumulating in fp32.  Tiles are staged
// global -> registers -> LDS with an XOR swizzle on the 16-byte slot
// (slot ^ ((row>>1)&7)) so that the ds_read_b128 fragment reads of a 16-lane
// group hit 16 distinct 4-bank slots (conflict free), double-buffered in LDS
// with the next tile's global loads in flight under the MFMAs (one barrier per
// K tile).  The MFMA is issued as D^T = W . A^T so each lane owns 4 consecutive
// output columns of one row -> 8-byte stores, and bias / activation / residual
// are applied in-register with the same bf16 rounding points as the reference's
// typed graph (oracle/ops.py).  Workgroup ids are remapped so that each XCD (own
// L2) walks a contiguous range of tiles.
#include <algorithm>

#include "common.hpp"
#include "internal.h"
#include "../../include/vlm_hip.h"

// gemm256_bf16.hip: the 256x256 phased kernel (-1: shape / epilogue not taken)
__attribute__((visibility("hidden"))) int vlm_gemm256_try(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K,
                    int lda, int ldw, int ldc, int ldres, int epilogue, void* stream);

__attribute__((visibility("hidden"))) void vlm_gemm256_set_variant(int v);
__attribute__((visibility("hidden"))) void vlm_gemm256_set_nf(int nf);
__attribute__((visibility("hidden"))) void vlm_gemm256_set_persist(int p);

namespace {

bool g_force_regstage = false;   // test hook (vlm_gemm_set_staging): exercise the register-staged kernel
int g_tile256 = 0;               // 0 = automatic, -1 = never use the 256x256 phased kernel, 1 = whenever it is legal

constexpr int BK = 64;          // k elements per LDS tile
constexpr int ROWB = BK * 2;    // bytes per tile row (128)

__device__ __forceinline__ int lds_off(int row, int slot) { return row * ROWB + ((slot ^ ((row >> 1) & 7)) << 4); }

// GLDS = true: tiles are staged with global_load_lds_dwordx4 (HBM -> LDS DMA, no VGPR round trip, no ds_write
// pass).  The DMA writes lane-linear (wave base + lane*16 B), so the XOR slot swizzle is applied on the per-lane
// SOURCE address and the LDS image is exactly the one the register-staged path builds.  Needs K % 64 == 0.
// PARTIAL (split-K, small M x N with a long K - the LLM down projection at prompt length): blockIdx.y selects a K range
// of kchunk elements; the workgroup writes its fp32 accumulators to part[split][M][N] (C reinterpreted) and
// splitk_reduce_kernel sums the splits in a fixed order and applies the epilogue.
// W4 = true (register-staged form only): W holds MLX affine 4-bit weights - uint32 words [N][K/8] (ldw = K) with
// Wsb uint32 [N][K/64] = (scale bf16 | bias bf16 << 16) per 64-wide group - and the W tile is DEQUANTISED ON ITS WAY
// INTO LDS: a thread's 16-byte LDS chunk (8 consecutive k of one row) is exactly one q word, and BK = 64 = the group
// size, so a K tile of a row has ONE (scale, bias).  The LDS image is bit for bit the one the bf16 kernel builds from
// vlm_dequant_w4's output (same fp32 scale * q + bias, one rounding: mx.dequantize), so the fused GEMM equals
// dequantise-then-GEMM bit for bit while reading 4.5 instead of 16 + 16 + 16 bits per weight (nn.QuantizedLinear /
// mx.quantized_matmul at L > 1, reference utils.py:918-967).
// LDS stages of the LDS-DMA K loop per tile shape (2 = the plain double buffer) and the counted wait it needs
template <int BM, int BN>
constexpr int gemm_stages() {
  if ((BM == 64 || BM == 32) && BN == 64) return 4;
#ifndef VLM_GEMM_NS128
#define VLM_GEMM_NS128 2      // measured (profiles/r06_wide_step.txt, item 5): 3 / 4 stages on the 128-wide tiles LOSE - one workgroup per CU instead of two
#endif
  if (BM == 128 && BN == 128) return VLM_GEMM_NS128;
  if (BM == 64 && BN == 128) return VLM_GEMM_NS128;
  return 2;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N == 6 || N == 8 || N == 12 || N == 16, "add the immediate");
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

template <int BM, int BN, int EPI, bool GLDS, bool PARTIAL = false, bool W4 = false>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                        const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                        bf16_t* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                                        int ldc, int ldres, int tiles_n, int nwg, int kchunk,
                                                        const unsigned* __restrict__ Wsb = nullptr) {
  static_assert(!W4 || !GLDS, "the 4-bit form stages W through registers");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ksplit = PARTIAL ? (int)blockIdx.y : 0;
  const unsigned* Wq = reinterpret_cast<const unsigned*>(W);      // (W4) q words, row pitch ldw / 8
  const int sb_pitch = ldw >> 6;                                  // (W4) groups per row
  if (PARTIAL) {
    A += (size_t)ksplit * kchunk;
    if (W4) { Wq += (size_t)ksplit * kchunk / 8; Wsb += (size_t)ksplit * kchunk / 64; }
    else W += (size_t)ksplit * kchunk;
    K = min(kchunk, K - ksplit * kchunk);
  }
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 16, NT = WN / 16;
  constexpr int A_PER = BM * 8 / 256, W_PER = BN * 8 / 256;
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;

  // XCD-aware bijective remap: blocks with the same (bid % 8) share an L2.
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // grouped order inside the XCD's range: the tiles an XCD runs at once (~64: 2 workgroups per CU) form an 8 x 8
  // block - 8 A row-blocks + 8 W row-blocks through its L2 instead of 1-2 + 64
  int pid_m, pid_n;
  {
    constexpr int GROUP_M = 8;
    const int tiles_m = nwg / tiles_n, per_group = GROUP_M * tiles_n, gid = bid / per_group;
    const int first_m = gid * GROUP_M, gsz = min(tiles_m - first_m, GROUP_M), r = bid - gid * per_group;
    pid_m = first_m + r % gsz;
    pid_n = r / gsz;
  }
  const int m0 = pid_m * BM, n0 = pid_n * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  uint4 ra[A_PER], rw[W_PER];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + 256 * i, row = c >> 3, slot = c & 7;
      const int gm = min(m0 + row, M - 1), gk = k0 + slot * 8;
      ra[i] = (gk < K) ? *reinterpret_cast<const uint4*>(A + (size_t)gm * lda + gk) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
      const int c = tid + 256 * i, row = c >> 3, slot = c & 7;
      const int gn = min(n0 + row, N - 1), gk = k0 + slot * 8;
      if constexpr (W4) {
        // raw words now (q word + the group's scale | bias); dequantised in lstore, after the MFMAs of the current tile
        rw[i] = (gk < K) ? make_uint4(Wq[(size_t)gn * (ldw >> 3) + (gk >> 3)], Wsb[(size_t)gn * sb_pitch + kt], 0, 0)
                         : make_uint4(0, 0, 0, 0);
      } else {
        rw[i] = (gk < K) ? *reinterpret_cast<const uint4*>(W + (size_t)gn * ldw + gk) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto lstore = [&](int buf) {
    char* as = smem + buf * STAGE;
    char* ws = as + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const int c = tid + 256 * i;
      *reinterpret_cast<uint4*>(as + lds_off(c >> 3, c & 7)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < W_PER; ++i) {
      const int c = tid + 256 * i;
      uint4 o = rw[i];
      if constexpr (W4) {
        const unsigned w = rw[i].x, sbw = rw[i].y;
        const float sc = bf_lo(sbw), bi = bf_hi(sbw);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __fadd_rn(__fmul_rn(sc, (float)((w >> (4 * j)) & 0xFu)), bi);    // == dequant_w4_kernel
        o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
      }
      *reinterpret_cast<uint4*>(ws + lds_off(c >> 3, c & 7)) = o;
    }
  };

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  auto compute = [&](int buf) {
    const char* as = smem + buf * STAGE;
    const char* ws = as + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t af[MT], wf[NT];
      const int slot = ks * 4 + (lane >> 4);
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int row = wm * WM + j * 16 + (lane & 15);
        af[j] = *reinterpret_cast<const bf16x8_t*>(as + lds_off(row, slot));
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int row = wn * WN + i * 16 + (lane & 15);
        wf[i] = *reinterpret_cast<const bf16x8_t*>(ws + lds_off(row, slot));
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
  };

  if (GLDS) {
    constexpr int A_IT = BM / 32, W_IT = BN / 32;   // 1 KiB (8 rows) DMA pieces per wave
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const int rl = lane >> 3, sp = lane & 7;
    const bf16_t* asrc[A_IT];
    const bf16_t* wsrc[W_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
      const int row = 8 * (uw * A_IT + j) + rl;
      asrc[j] = A + (size_t)min(m0 + row, M - 1) * lda + ((sp ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int j = 0; j < W_IT; ++j) {
      const int row = 8 * (uw * W_IT + j) + rl;
      wsrc[j] = W + (size_t)min(n0 + row, N - 1) * ldw + ((sp ^ ((row >> 1) & 7)) << 3);
    }
    auto issue = [&](int kt, int buf) {
      char* as = smem + buf * STAGE;
      char* ws = as + A_BYTES;
#pragma unroll
      for (int j = 0; j < A_IT; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + (size_t)kt * BK),
                                         (__attribute__((address_space(3))) void*)(as + (uw * A_IT + j) * 1024), 16, 0, 0);
#pragma unroll
      for (int j = 0; j < W_IT; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[j] + (size_t)kt * BK),
                                         (__attribute__((address_space(3))) void*)(ws + (uw * W_IT + j) * 1024), 16, 0, 0);
    };
    constexpr int NS = gemm_stages<BM, BN>();
    if constexpr (NS > 2) {
      // 64x64 tiles carry ~100 cycles of MFMA per K tile: with one tile of prefetch every iteration waited a full
      // memory latency (~0.8 us per K tile measured on the M = 386 prefill GEMMs).  NS stages, NS - 1 tiles in
      // flight, COUNTED waits (PER DMA instructions per wave and tile; tile indices clamped at the end so the count
      // never changes), raw barriers (__syncthreads would drain vmcnt(0)).  Stage kt % NS is refilled with tile kt + NS
      // in iteration kt + 1, after the barrier that ends its last read.
      // (round 6) BM = 32, for GEMMs of <= 32 rows: half the A tile, 12 KiB stages - three workgroups per CU instead of two
      // (a weight stream wants bytes in flight) and 3 DMA instructions per wave and tile.
      // (round 6) the loop is generic in the stage count; the 128-wide tiles were tried at 3 and 4 stages (-DVLM_GEMM_NS128=3 / 4)
      // and keep their double buffer: the deeper pipeline costs the second workgroup of the CU (fc1 at M = 1024: 24.8 -> 33 us).
      constexpr int PER = A_IT + W_IT, WAITN = (NS - 2) * PER;
#pragma unroll
      for (int t = 0; t < NS - 1; ++t) issue(min(t, nk - 1), t);
      wait_vmcnt<WAITN>();                                  // tile 0 landed (tiles 1 .. NS - 2 may be in flight)
      __builtin_amdgcn_s_barrier();
      for (int kt = 0; kt < nk; ++kt) {
        issue(min(kt + NS - 1, nk - 1), (kt + NS - 1) % NS);
        __builtin_amdgcn_sched_barrier(0);
        compute(kt % NS);
        __builtin_amdgcn_sched_barrier(0);
        wait_vmcnt<WAITN>();                                // tile kt + 1 landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                       // ... and everybody's; stage kt % NS is free again
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // clamped reloads still target LDS
      __builtin_amdgcn_s_barrier();
    } else {
      issue(0, 0);
      __syncthreads();   // hipcc drains vmcnt(0) for the in-flight LDS DMA before the barrier
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        compute(kt & 1);
        __syncthreads();
      }
    }
  } else {
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = (kt + 1) < nk;
      if (more) gload(kt + 1);
      compute(kt & 1);
      if (more) lstore((kt + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue.  Lane holds D^T[n = nb + (lane>>4)*4 + r][m = mb + (lane&15)].  Bias / activation are applied
  //      in registers (rounded to bf16 at the reference's points), the tile is transposed through LDS (free after
  //      the k loop) and written with fully coalesced 16-byte stores; the residual is added in that pass from
  //      equally coalesced 16-byte loads.
  if (PARTIAL) {
    float* part = reinterpret_cast<float*>(C) + (size_t)ksplit * M * N;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + wm * WM + j * 16 + (lane & 15);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int n = n0 + wn * WN + i * 16 + (lane >> 4) * 4;
        if (m < M && n < N)   // N % 8 == 0 and n % 4 == 0: the float4 is inside the row
          *reinterpret_cast<float4*>(part + (size_t)m * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    return;
  }
  constexpr bool SWI = (EPI & VLM_EPI_SWIGLU) != 0;
  constexpr int OUT_N = SWI ? BN / 2 : BN;      // output columns of this tile
  constexpr int C_LD = OUT_N + 8;               // padded row (elements)
  bf16_t* cs = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int ml = wm * WM + j * 16 + (lane & 15);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int nl = wn * WN + i * 16 + (lane >> 4) * 4;
      const int n = min(n0 + nl, N - 4);
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (EPI & VLM_EPI_BIAS) {
        const uint2 b = *reinterpret_cast<const uint2*>(bias + n);
        v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
      }
      if (SWI) {
        // interleaved (gate, up) rows of W -> N/2 outputs
        const float o0 = swiglu_(rbf(v[0]), rbf(v[1])), o1 = swiglu_(rbf(v[2]), rbf(v[3]));
        *reinterpret_cast<uint32_t*>(cs + ml * C_LD + (nl >> 1)) = pack_bf2(o0, o1);
      } else {
        if (EPI & VLM_EPI_GELU_FAST) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_fast_(rbf(v[r]));
        }
        if (EPI & VLM_EPI_GELU_ERF) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_erf_(rbf(v[r]));
        }
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(cs + ml * C_LD + nl) = o;
      }
    }
  }
  __syncthreads();
  {
    constexpr int CPR = OUT_N / 8;                // 16-byte chunks per tile row
    const int n_out = SWI ? (N >> 1) : N, n0o = SWI ? (n0 >> 1) : n0;
#pragma unroll 2
    for (int c = tid; c < BM * CPR; c += 256) {
      const int row = c / CPR, cc = c % CPR;
      const int m = m0 + row, n = n0o + cc * 8;
      if (m < M && n < n_out) {
        uint4 u = *reinterpret_cast<const uint4*>(cs + row * C_LD + cc * 8);
        if (EPI & VLM_EPI_ROPE2D) u = vlm_rope2d_chunk(u, res, M, m, n, ldres);
        if (EPI & VLM_EPI_RESIDUAL) {
          const uint4 r = *reinterpret_cast<const uint4*>(res + (size_t)m * ldres + n);
          u.x = pack_bf2(bf_lo(u.x) + bf_lo(r.x), bf_hi(u.x) + bf_hi(r.x));
          u.y = pack_bf2(bf_lo(u.y) + bf_lo(r.y), bf_hi(u.y) + bf_hi(r.y));
          u.z = pack_bf2(bf_lo(u.z) + bf_lo(r.z), bf_hi(u.z) + bf_hi(r.z));
          u.w = pack_bf2(bf_lo(u.w) + bf_lo(r.w), bf_hi(u.w) + bf_hi(r.w));
        }
        *reinterpret_cast<uint4*>(C + (size_t)m * ldc + n) = u;
      }
    }
  }
}

// sum of the split-K partials (fixed order: deterministic) + the epilogue of gemm_bf16_kernel, 8 columns per thread
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits,
                                                            const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                            bf16_t* __restrict__ C, int M, int N, int ldc, int ldres) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x, per_row = N >> 3;
  if (idx >= (long)M * per_row) return;
  const int m = (int)(idx / per_row), n = (int)(idx % per_row) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < splits; ++s) {
    const float4* p = reinterpret_cast<const float4*>(part + ((size_t)s * M + m) * N + n);
    const float4 a = p[0], b = p[1];
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  if (EPI & VLM_EPI_BIAS) {
    const uint4 b = *reinterpret_cast<const uint4*>(bias + n);
    v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
    v[4] += bf_lo(b.z); v[5] += bf_hi(b.z); v[6] += bf_lo(b.w); v[7] += bf_hi(b.w);
  }
  if (EPI & VLM_EPI_GELU_FAST) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = gelu_fast_(rbf(v[r]));
  }
  if (EPI & VLM_EPI_GELU_ERF) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = gelu_erf_(rbf(v[r]));
  }
  uint4 u;
  u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
  if (EPI & VLM_EPI_RESIDUAL) {
    const uint4 r = *reinterpret_cast<const uint4*>(res + (size_t)m * ldres + n);
    u.x = pack_bf2(bf_lo(u.x) + bf_lo(r.x), bf_hi(u.x) + bf_hi(r.x));
    u.y = pack_bf2(bf_lo(u.y) + bf_lo(r.y), bf_hi(u.y) + bf_hi(r.y));
    u.z = pack_bf2(bf_lo(u.z) + bf_lo(r.z), bf_hi(u.z) + bf_hi(r.z));
    u.w = pack_bf2(bf_lo(u.w) + bf_lo(r.w), bf_hi(u.w) + bf_hi(r.w));
  }
  *reinterpret_cast<uint4*>(C + (size_t)m * ldc + n) = u;
}

// the fp32 partials of TWO 8-wide chunks (row m, columns n0.. and n1..) summed over the splits in the fixed order 0 .. splits - 1
// (splitk_reduce_kernel's sum), with every load of both chunks issued before the first add: up to 8 splits x 2 chunks x 32 bytes
// in flight per thread instead of one dependent round trip per split
__device__ __forceinline__ void reduce_two_chunks(const float* __restrict__ part, int splits, int M, int N, int m, int n0, int n1,
                                                  float (&v0)[8], float (&v1)[8]) {
  // (ext_vector loads from a CLAMPED split index, the absent splits' values replaced by +0 afterwards: written with float4 and
  // a conditional load, hipcc emitted one dword load + branch per element)
  f32x4_t a0[8], b0[8], a1[8], b1[8];
#pragma unroll
  for (int sp = 0; sp < 8; ++sp) {
    const size_t base = ((size_t)min(sp, splits - 1) * M + m) * N;
    const f32x4_t* p0 = reinterpret_cast<const f32x4_t*>(part + base + n0);
    const f32x4_t* p1 = reinterpret_cast<const f32x4_t*>(part + base + n1);
    a0[sp] = p0[0]; b0[sp] = p0[1];
    a1[sp] = p1[0]; b1[sp] = p1[1];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { v0[j] = 0.f; v1[j] = 0.f; }
#pragma unroll
  for (int sp = 0; sp < 8; ++sp) {
    if (sp < splits) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { v0[j] += a0[sp][j]; v0[4 + j] += b0[sp][j]; v1[j] += a1[sp][j]; v1[4 + j] += b1[sp][j]; }
    }
  }
  for (int sp = 8; sp < splits; ++sp) {                    // (only the forced-split test hook goes past 8)
    const f32x4_t* p0 = reinterpret_cast<const f32x4_t*>(part + ((size_t)sp * M + m) * N + n0);
    const f32x4_t* p1 = reinterpret_cast<const f32x4_t*>(part + ((size_t)sp * M + m) * N + n1);
    const f32x4_t a = p0[0], bq = p0[1], c = p1[0], d = p1[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v0[j] += a[j]; v0[4 + j] += bq[j]; v1[j] += c[j]; v1[4 + j] += d[j]; }
  }
}

// ---- split-K reduce launches that also do the NEXT launch's work (round 6: the wide decode steps, csrc/engine.hip).  A layer of a
// 17..64-row step is the prefill's launch sequence; its three small GEMMs (qkv, o_proj, down) are split-K, so each is already
// followed by a reduce launch - and then by a launch that reads the reduced rows straight back: RMSNorm (after o_proj and after
// down) or M-RoPE + KV write (after qkv).  These kernels are the reduce AND that follower, arithmetic and order of both kept
// (the reduce's fixed split order, bias, one rounding to bf16, residual in bf16; then the follower on the rounded values), so
// the rows they write equal the two-launch sequence bit for bit (tests/test_ops_gpu.py).  6.7 + 6.7 + 6.3 us of 152 per 7B layer.

// reduce + epilogue -> C, then y = w * T(h * rsqrt(mean(h^2) + eps)) of the row just written (rmsnorm_kernel of norm.hip: one
// wave per row there, lane -> chunks lane + 64 i, one running sum of squares in (i, j) order, wave_sum).  Here a row has the whole
// workgroup: wave w reduces chunks i = w, w + 4 (the loads are the cost), the rounded row meets in LDS, then EVERY wave takes the
// sum of squares over ALL chunks in that same order - the same inv in all four - and normalises its own chunks.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_norm_kernel(const float* __restrict__ part, int splits,
                                                                 const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                                 bf16_t* __restrict__ C, int M, int N, int ldc, int ldres,
                                                                 const bf16_t* __restrict__ norm_w, float eps,
                                                                 bf16_t* __restrict__ xn, int ldxn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* rowbuf = reinterpret_cast<uint4*>(smem);                 // the row as written to C: N / 8 chunks of 8 bf16
  const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nchunk = N >> 3;
  // this wave's chunks: i = wave and wave + 4 (N <= 4096: at most two); a lane past the row's end reduces chunk 0 again, unused
  const int c0 = lane + wave * 64, c1 = lane + (wave + 4) * 64;
  const bool in0 = c0 < nchunk, in1 = c1 < nchunk;
  uint4 mine[2];
  {
    float v[2][8];
    reduce_two_chunks(part, splits, M, N, m, in0 ? c0 * 8 : 0, in1 ? c1 * 8 : 0, v[0], v[1]);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = q ? c1 : c0, n = c * 8;
      if (!(q ? in1 : in0)) continue;
      if (EPI & VLM_EPI_BIAS) {
        const uint4 bb = *reinterpret_cast<const uint4*>(bias + n);
        v[q][0] += bf_lo(bb.x); v[q][1] += bf_hi(bb.x); v[q][2] += bf_lo(bb.y); v[q][3] += bf_hi(bb.y);
        v[q][4] += bf_lo(bb.z); v[q][5] += bf_hi(bb.z); v[q][6] += bf_lo(bb.w); v[q][7] += bf_hi(bb.w);
      }
      uint4 u;
      u.x = pack_bf2(v[q][0], v[q][1]); u.y = pack_bf2(v[q][2], v[q][3]); u.z = pack_bf2(v[q][4], v[q][5]); u.w = pack_bf2(v[q][6], v[q][7]);
      if (EPI & VLM_EPI_RESIDUAL) {
        const uint4 r = *reinterpret_cast<const uint4*>(res + (size_t)m * ldres + n);
        u.x = pack_bf2(bf_lo(u.x) + bf_lo(r.x), bf_hi(u.x) + bf_hi(r.x));
        u.y = pack_bf2(bf_lo(u.y) + bf_lo(r.y), bf_hi(u.y) + bf_hi(r.y));
        u.z = pack_bf2(bf_lo(u.z) + bf_lo(r.z), bf_hi(u.z) + bf_hi(r.z));
        u.w = pack_bf2(bf_lo(u.w) + bf_lo(r.w), bf_hi(u.w) + bf_hi(r.w));
      }
      *reinterpret_cast<uint4*>(C + (size_t)m * ldc + n) = u;
      rowbuf[c] = u;
      mine[q] = u;
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i * 64 < nchunk; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      const uint4 u = rowbuf[c];
      const float f[8] = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y), bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[j] * f[j];
    }
  }
  const float inv = rsqrtf(wave_sum(s) / (float)N + eps);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = q ? c1 : c0;
    if (!(q ? in1 : in0)) continue;
    const uint4 u = mine[q], wu = reinterpret_cast<const uint4*>(norm_w)[c];
    const float f[8] = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y), bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
    const float wv[8] = {bf_lo(wu.x), bf_hi(wu.x), bf_lo(wu.y), bf_hi(wu.y), bf_lo(wu.z), bf_hi(wu.z), bf_lo(wu.w), bf_hi(wu.w)};
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = wv[j] * rbf(f[j] * inv);
    uint4 o;
    o.x = pack_bf2(r[0], r[1]); o.y = pack_bf2(r[2], r[3]); o.z = pack_bf2(r[4], r[5]); o.w = pack_bf2(r[6], r[7]);
    *reinterpret_cast<uint4*>(xn + (size_t)m * ldxn + c * 8) = o;
  }
}

// reduce + bias -> the qkv rows, rotated (q, k) and written to the paged cache (k, v): mrope_kvwrite_kernel of rope.hip in its decode
// form (row b = one token at text position pos[b], slot[b] of block-table row b) on the values the reduce would have stored.
// One thread = one 8-wide chunk of the low half of a q / k head + its partner in the high half, or one chunk of a v head.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_rope_kernel(const float* __restrict__ part, int splits,
                                                                 const bf16_t* __restrict__ bias, bf16_t* __restrict__ C, int M,
                                                                 int N, int ldc, const VlmGemmTail t) {
  const int D = t.D, Hq = t.Hq, Hkv = t.Hkv, half = D >> 1, cph = half >> 3;
  const float* inv_freq = t.inv_freq;
  if (t.long_from > 0) {
    bool any_long = false;
    for (int r = 0; r < M; ++r) any_long |= t.slot[r] >= t.long_from;
    if (any_long) inv_freq += half;
  }
  const int rot_items = (Hq + Hkv) * cph, v_items = Hkv * (D >> 3), per_tok = rot_items + v_items;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * per_tok) return;
  const int tok = (int)(idx / per_tok), it = (int)(idx % per_tok);
  auto finish = [&](float (&v)[8], int n) -> uint4 {       // what splitk_reduce_kernel<EPI> stores at (tok, n .. n + 7)
    if (EPI & VLM_EPI_BIAS) {
      const uint4 b = *reinterpret_cast<const uint4*>(bias + n);
      v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
      v[4] += bf_lo(b.z); v[5] += bf_hi(b.z); v[6] += bf_lo(b.w); v[7] += bf_hi(b.w);
    }
    uint4 u;
    u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
    return u;
  };
  bf16_t* row = C + (size_t)tok * ldc;
  const int slot = t.slot[tok];
  const long page = t.block_table[(size_t)tok * t.max_pages + (slot >> 6)];
  const int within = slot & 63;
  if (it < rot_items) {
    const int head = it / cph, c = it % cph;
    const int n = head * D + c * 8;
    float va[8], vb[8];
    reduce_two_chunks(part, splits, M, N, tok, n, n + half, va, vb);
    const uint4 ua = finish(va, n), ub = finish(vb, n + half);
    const float a[8] = {bf_lo(ua.x), bf_hi(ua.x), bf_lo(ua.y), bf_hi(ua.y), bf_lo(ua.z), bf_hi(ua.z), bf_lo(ua.w), bf_hi(ua.w)};
    const float b[8] = {bf_lo(ub.x), bf_hi(ub.x), bf_lo(ub.y), bf_hi(ub.y), bf_lo(ub.z), bf_hi(ub.z), bf_lo(ub.w), bf_hi(ub.w)};
    float oa[8], ob[8];
    const float p = (float)t.pos[tok];                     // decode rows: all three M-RoPE axes at the text position
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = c * 8 + j;
      const float ang = p * inv_freq[f];
      float sn, co;
      sincosf(ang, &sn, &co);
      const float za = rbf(a[j] * t.qk_scale), zb = rbf(b[j] * t.qk_scale);
      oa[j] = za * co - zb * sn;
      ob[j] = zb * co + za * sn;
    }
    uint4 lo, hi;
    lo.x = pack_bf2(oa[0], oa[1]); lo.y = pack_bf2(oa[2], oa[3]); lo.z = pack_bf2(oa[4], oa[5]); lo.w = pack_bf2(oa[6], oa[7]);
    hi.x = pack_bf2(ob[0], ob[1]); hi.y = pack_bf2(ob[2], ob[3]); hi.z = pack_bf2(ob[4], ob[5]); hi.w = pack_bf2(ob[6], ob[7]);
    *reinterpret_cast<uint4*>(row + n) = lo;
    *reinterpret_cast<uint4*>(row + n + half) = hi;
    if (head >= Hq) {
      const int g = head - Hq;
      const size_t kb = ((size_t)page * Hkv + g) * (size_t)(D >> 3);
      *reinterpret_cast<uint4*>(t.kpool + ((kb + c) * 64 + within) * 8) = lo;
      *reinterpret_cast<uint4*>(t.kpool + ((kb + c + cph) * 64 + within) * 8) = hi;
    }
  } else {
    const int vi = it - rot_items;
    const int g = vi / (D >> 3), c = vi % (D >> 3);
    const int n = (Hq + Hkv + g) * D + c * 8;
    float va[8], vdup[8];
    reduce_two_chunks(part, splits, M, N, tok, n, n, va, vdup);      // (one chunk: the second slot re-reads it)
    const uint4 v = finish(va, n);
    *reinterpret_cast<uint4*>(row + n) = v;
    bf16_t* vb = t.vpool + (((size_t)page * Hkv + g) * D + c * 8) * 64 + vlm_vslot(within);   // [D][64 slots]
    vb[0 * 64] = (bf16_t)(v.x & 0xffffu); vb[1 * 64] = (bf16_t)(v.x >> 16);
    vb[2 * 64] = (bf16_t)(v.y & 0xffffu); vb[3 * 64] = (bf16_t)(v.y >> 16);
    vb[4 * 64] = (bf16_t)(v.z & 0xffffu); vb[5 * 64] = (bf16_t)(v.z >> 16);
    vb[6 * 64] = (bf16_t)(v.w & 0xffffu); vb[7 * 64] = (bf16_t)(v.w >> 16);
  }
}

// The tail request of the calling thread's next split-K GEMM (vlm_gemm_bf16_tail / vlm_gemm_w4_tail set and clear it around
// gemm_dispatch: the templates between the entry point and the reduce launch stay as they are).
thread_local const VlmGemmTail* t_tail = nullptr;
thread_local bool t_tail_done = false;

// the reduce launch of a split-K GEMM: the plain one, or - when the caller asked for a tail this launch can carry - the fused one
template <int EPI>
void launch_reduce(const float* ws, int splits, const void* bias, const void* res, void* C, int M, int N, int ldc, int ldres,
                   hipStream_t st) {
  const VlmGemmTail* t = t_tail;
  if (t && !(EPI & (VLM_EPI_GELU_FAST | VLM_EPI_GELU_ERF | VLM_EPI_SWIGLU | VLM_EPI_ROPE2D))) {
    if (t->kind == VLM_TAIL_RMSNORM && N <= 4096 && t->norm_w && t->xn && t->ldxn % 8 == 0) {
      hipLaunchKernelGGL((splitk_reduce_norm_kernel<EPI>), dim3(M), dim3(256), (size_t)N * 2, st, ws, splits, (const bf16_t*)bias,
                         (const bf16_t*)res, (bf16_t*)C, M, N, ldc, ldres, (const bf16_t*)t->norm_w, t->eps, (bf16_t*)t->xn, t->ldxn);
      t_tail_done = true;
      return;
    }
    if (t->kind == VLM_TAIL_ROPE_KV && !(EPI & VLM_EPI_RESIDUAL) && M <= 64 && t->D % 16 == 0 && N == (t->Hq + 2 * t->Hkv) * t->D) {
      const long total = (long)M * ((t->Hq + t->Hkv) * (t->D / 16) + t->Hkv * (t->D / 8));
      hipLaunchKernelGGL((splitk_reduce_rope_kernel<EPI>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, splits,
                         (const bf16_t*)bias, (bf16_t*)C, M, N, ldc, *t);
      t_tail_done = true;
      return;
    }
  }
  const long items = (long)M * (N >> 3);
  hipLaunchKernelGGL((splitk_reduce_kernel<EPI>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, ws, splits,
                     (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)C, M, N, ldc, ldres);
}

// per-process fp32 workspace of the split-K path (grown on demand; never (re)allocated while the stream is capturing)
// Split-K partials: one workspace PER STREAM (a prefill on a side stream may run under another one), sized once for
// everything the automatic policy can ask for: split-K is taken only when M*N < 256 tiles of 64x64 = 2^20 outputs and
// splits <= 8, i.e. <= 32 MiB of fp32 partials.  No growth, hence no device-wide synchronisation in the launch path
// (only the forced-split test hook can exceed the size and re-allocates).
struct SplitkWs {
  hipStream_t st;
  float* p;
  size_t bytes;
  bool borrowed;      // an alias of another stream's workspace (vlm_gemm_splitk_share): never freed through this entry
};
constexpr size_t SPLITK_WS_BYTES = 34u << 20;
constexpr int MAX_SPLITK_WS = 8;
SplitkWs g_splitk_ws[MAX_SPLITK_WS];
int g_n_splitk_ws = 0;
const bool g_midsize = [] { const char* e = getenv("VLM_GEMM_MIDSIZE"); return !e || atoi(e) != 0; }();   // A/B knob: 0 = the mid-size tile / split policy of rounds 1-5
const bool g_skinny32 = [] { const char* e = getenv("VLM_GEMM_SKINNY32"); return !e || atoi(e) != 0; }();   // A/B knob: 0 = 64-row A tiles for <= 32 rows too
const bool g_skinny64 = [] { const char* e = getenv("VLM_GEMM_SKINNY64"); return !e || atoi(e) != 0; }();   // A/B knob: 0 = the tile policy of rounds 1-5
int g_force_cfg = 0;   // tile of the plain kernels forced (vlm_gemm_set_staging mode 100 + 10 * splits + cfg): 1 = 64 x 64, 2 = 64 x 128, 3 = 128 x 128
int g_splitk = 0;   // 0 = automatic, -1 = never (vlm_gemm_set_staging mode 8), n > 1 = forced split count (test hook, 9: 4)

float* splitk_workspace(size_t bytes, hipStream_t st) {
  SplitkWs* slot = nullptr;
  for (int i = 0; i < g_n_splitk_ws; ++i)
    if (g_splitk_ws[i].st == st) slot = &g_splitk_ws[i];
  if (slot && bytes <= slot->bytes) return slot->p;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
  if (!slot) {
    if (g_n_splitk_ws == MAX_SPLITK_WS) return nullptr;      // caller falls back to the single-pass kernels
    slot = &g_splitk_ws[g_n_splitk_ws];
    *slot = SplitkWs{st, nullptr, 0, false};
  } else {
    if (slot->borrowed) return nullptr;
    if (hipStreamSynchronize(st) != hipSuccess) return nullptr;   // forced-split hook only: nobody may still read the old one
    (void)hipFree(slot->p);
    slot->p = nullptr;
    slot->bytes = 0;
  }
  const size_t want = bytes > SPLITK_WS_BYTES ? bytes + (bytes >> 1) : SPLITK_WS_BYTES;
  if (hipMalloc(reinterpret_cast<void**>(&slot->p), want) != hipSuccess) return nullptr;
  slot->bytes = want;
  if (slot == &g_splitk_ws[g_n_splitk_ws]) ++g_n_splitk_ws;
  return slot->p;
}

}  // namespace

// A graph is captured on a stream of its own, where nothing may be allocated - but its replays are ordered on the LAUNCH
// stream, so the captured split-K GEMMs may use that stream's workspace: `to` borrows the workspace of `from` (allocated
// here if need be) until vlm_gemm_splitk_unshare(to).  (Without it a captured GEMM falls back to the single-pass kernels:
// a wide decode step's down projection - 56 tiles x K = 18944 - on 56 workgroups.)
VLM_INTERNAL int vlm_gemm_splitk_share(void* from, void* to) {
  float* p = splitk_workspace(1, (hipStream_t)from);
  if (!p) return VLM_ERR_HIP;
  size_t bytes = 0;
  for (int i = 0; i < g_n_splitk_ws; ++i)
    if (g_splitk_ws[i].st == (hipStream_t)from) bytes = g_splitk_ws[i].bytes;
  for (int i = 0; i < g_n_splitk_ws; ++i)
    if (g_splitk_ws[i].st == (hipStream_t)to) return g_splitk_ws[i].borrowed ? VLM_OK : VLM_ERR_ARG;
  if (g_n_splitk_ws == MAX_SPLITK_WS) return VLM_ERR_ARG;
  g_splitk_ws[g_n_splitk_ws++] = SplitkWs{(hipStream_t)to, p, bytes, true};
  return VLM_OK;
}

VLM_INTERNAL void vlm_gemm_splitk_unshare(void* to) {
  for (int i = 0; i < g_n_splitk_ws; ++i)
    if (g_splitk_ws[i].st == (hipStream_t)to && g_splitk_ws[i].borrowed) {
      g_splitk_ws[i] = g_splitk_ws[g_n_splitk_ws - 1];
      --g_n_splitk_ws;
      return;
    }
}

namespace {

template <int EPI>
int launch_splitk(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
                  int ldw, int ldc, int ldres, int splits, float* ws, hipStream_t st) {
  const int kchunk = vlm_cdiv(vlm_cdiv(K, splits), BK) * BK;
  splits = vlm_cdiv(K, kchunk);
  if (M <= 32 && g_skinny32) {                       // (<= 32 rows: 32 x 64 tiles, see launch_epi)
    const int tiles_n = vlm_cdiv(N, 64), nwg = tiles_n;
    hipLaunchKernelGGL((gemm_bf16_kernel<32, 64, VLM_EPI_NONE, true, true>), dim3(nwg, splits), dim3(256), 4 * (size_t)(32 + 64) * ROWB, st,
                       (const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                       reinterpret_cast<bf16_t*>(ws), M, N, K, lda, ldw, N, 0, tiles_n, nwg, kchunk);
    launch_reduce<EPI>((const float*)ws, splits, bias, res, C, M, N, ldc, ldres, st);
    hipError_t e32 = hipGetLastError();
    return e32 == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e32;
  }
  const int tiles_m = vlm_cdiv(M, 64), tiles_n = vlm_cdiv(N, 64), nwg = tiles_m * tiles_n;
  const size_t lds = 4 * (size_t)(64 + 64) * ROWB;   // four stages (see the 64x64 K loop)
  hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, VLM_EPI_NONE, true, true>), dim3(nwg, splits), dim3(256), lds, st,
                     (const bf16_t*)A, (const bf16_t*)W, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                     reinterpret_cast<bf16_t*>(ws), M, N, K, lda, ldw, N, 0, tiles_n, nwg, kchunk);
  launch_reduce<EPI>((const float*)ws, splits, bias, res, C, M, N, ldc, ldres, st);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

template <int BM, int BN, int EPI, bool GLDS>
int launch_cfg(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
               int ldw, int ldc, int ldres, hipStream_t st) {
  const int tiles_m = vlm_cdiv(M, BM), tiles_n = vlm_cdiv(N, BN), nwg = tiles_m * tiles_n;
  const size_t lds = (GLDS ? gemm_stages<BM, BN>() : 2) * (size_t)(BM + BN) * ROWB;
  if (lds > 65536) {          // beyond the default dynamic-LDS limit: raised once per instantiation
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, EPI, GLDS>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return VLM_ERR_HIP + (int)attr;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, EPI, GLDS>), dim3(nwg), dim3(256), lds, st, (const bf16_t*)A, (const bf16_t*)W,
                     (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)C, M, N, K, lda, ldw, ldc, ldres, tiles_n, nwg, 0);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

// ---- MLX 4-bit W (dequant-fused form): register-staged kernels only
template <int EPI>
int launch_splitk_w4(const void* A, const void* Wq, const void* Wsb, const void* bias, const void* res, void* C, int M, int N,
                     int K, int lda, int ldc, int ldres, int splits, float* ws, hipStream_t st) {
  const int kchunk = vlm_cdiv(vlm_cdiv(K, splits), BK) * BK;
  splits = vlm_cdiv(K, kchunk);
  const int tiles_m = vlm_cdiv(M, 64), tiles_n = vlm_cdiv(N, 64), nwg = tiles_m * tiles_n;
  const size_t lds = 2 * (size_t)(64 + 64) * ROWB;
  hipLaunchKernelGGL((gemm_bf16_kernel<64, 64, VLM_EPI_NONE, false, true, true>), dim3(nwg, splits), dim3(256), lds, st,
                     (const bf16_t*)A, (const bf16_t*)Wq, (const bf16_t*)nullptr, (const bf16_t*)nullptr,
                     reinterpret_cast<bf16_t*>(ws), M, N, K, lda, K, N, 0, tiles_n, nwg, kchunk, (const unsigned*)Wsb);
  launch_reduce<EPI>((const float*)ws, splits, bias, res, C, M, N, ldc, ldres, st);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

template <int BM, int BN, int EPI>
int launch_cfg_w4(const void* A, const void* Wq, const void* Wsb, const void* bias, const void* res, void* C, int M, int N, int K,
                  int lda, int ldc, int ldres, hipStream_t st) {
  const int tiles_m = vlm_cdiv(M, BM), tiles_n = vlm_cdiv(N, BN), nwg = tiles_m * tiles_n;
  const size_t lds = 2 * (size_t)(BM + BN) * ROWB;
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, EPI, false, false, true>), dim3(nwg), dim3(256), lds, st, (const bf16_t*)A,
                     (const bf16_t*)Wq, (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)C, M, N, K, lda, K, ldc, ldres, tiles_n,
                     nwg, 0, (const unsigned*)Wsb);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

template <int EPI>
int launch_epi_w4(const void* A, const void* Wq, const void* Wsb, const void* bias, const void* res, void* C, int M, int N, int K,
                  int lda, int ldc, int ldres, hipStream_t st) {
  const long t128 = (long)vlm_cdiv(M, 128) * vlm_cdiv(N, 128);
  const long t64n = (long)vlm_cdiv(M, 64) * vlm_cdiv(N, 128);
  if (!(EPI & VLM_EPI_SWIGLU) && g_splitk >= 0 && N % 8 == 0) {          // same split-K policy as the bf16 form
    const long t64 = (long)vlm_cdiv(M, 64) * vlm_cdiv(N, 64);
    int splits = g_splitk > 1 ? g_splitk : 0;
    if (!splits && t64 < 256 && K >= 2048) splits = (int)std::min<long>(8, std::max<long>(2, (512 + t64 - 1) / t64));
    if (!splits && t64 < 256 && M <= 64 && g_skinny64 && K >= 1024) {        // (see launch_epi)
      splits = (int)std::min<long>(8, std::max<long>(2, (512 + t64 - 1) / t64));
      while (splits > 1 && K / splits < 4 * BK) --splits;
    }
    if (splits > 1 && K / splits >= 4 * BK) {
      if (float* ws = splitk_workspace((size_t)splits * M * N * sizeof(float), st))
        return launch_splitk_w4<EPI>(A, Wq, Wsb, bias, res, C, M, N, K, lda, ldc, ldres, splits, ws, st);
    }
  }
  if (M <= 64 && g_skinny64) return launch_cfg_w4<64, 64, EPI>(A, Wq, Wsb, bias, res, C, M, N, K, lda, ldc, ldres, st);   // (see launch_epi)
  if (t128 >= 200) return launch_cfg_w4<128, 128, EPI>(A, Wq, Wsb, bias, res, C, M, N, K, lda, ldc, ldres, st);
  if (t64n >= 200) return launch_cfg_w4<64, 128, EPI>(A, Wq, Wsb, bias, res, C, M, N, K, lda, ldc, ldres, st);
  return launch_cfg_w4<64, 64, EPI>(A, Wq, Wsb, bias, res, C, M, N, K, lda, ldc, ldres, st);
}

template <int EPI>
int launch_epi(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
               int ldw, int ldc, int ldres, hipStream_t st) {
  // pick the largest tile that still yields >= ~1 workgroup per CU (256 CUs); LDS-DMA staging when K % 64 == 0
  const long t128 = (long)vlm_cdiv(M, 128) * vlm_cdiv(N, 128);
  const long t64n = (long)vlm_cdiv(M, 64) * vlm_cdiv(N, 128);
  const bool glds = (K % BK == 0) && !g_force_regstage;
  // split-K: few 64x64 tiles and a long K (LLM down projection at prompt length: 168 tiles x 140 K tiles, measured
  // 106 TF) -> enough (tile, K range) workgroups to fill the chip, fp32 partials summed by splitk_reduce_kernel
  if (glds && !(EPI & (VLM_EPI_SWIGLU | VLM_EPI_ROPE2D)) && g_splitk >= 0 && N % 8 == 0) {
    const long t64 = (long)vlm_cdiv(M, 64) * vlm_cdiv(N, 64);
    int splits = g_splitk > 1 ? g_splitk : 0;
    if (!splits && t64 < 256 && K >= 2048) splits = (int)std::min<long>(8, std::max<long>(2, (512 + t64 - 1) / t64));
    // ... and up to 64 rows (a weight stream: the 2B model's qkv / o_proj at a 64-row step are 32 / 24 tiles of K = 1536, 10 us
    // each for 6 / 5 MB) from K = 1024, with as many splits as leave 4 K tiles per workgroup
    if (!splits && t64 < 256 && M <= 64 && g_skinny64 && K >= 1024) {
      splits = (int)std::min<long>(8, std::max<long>(2, (512 + t64 - 1) / t64));
      while (splits > 1 && K / splits < 4 * BK) --splits;
    }
    // ... and 256 .. 640 tiles over a VERY long K (the 7B down projection of a prompt of a few hundred tokens: 392 tiles x 296 K
    // tiles): three K ranges - T = 386: 137 -> 125 us, T = 640: 206 -> 179 (profiles/r06_wide_step.txt, item 6).  At K = 14336
    // and below the partials cost what the split gains.
    if (!splits && g_midsize && t64 >= 256 && t64 < 640 && K >= 16384) splits = 3;
    if (splits > 1 && K / splits >= 4 * BK) {
      if (float* ws = splitk_workspace((size_t)splits * M * N * sizeof(float), st))
        return launch_splitk<EPI>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, splits, ws, st);
    }
  }
#define CFG(BMV, BNV)                                                                                            \
  (glds ? launch_cfg<BMV, BNV, EPI, true>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st)                \
        : launch_cfg<BMV, BNV, EPI, false>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st))
  // up to 64 rows (the wide decode steps, short prompts) the launch is a WEIGHT STREAM: what counts is bytes in flight, i.e.
  // workgroups - 7B gate/up at 32 rows: 148 tiles of 256 rows 65.9 us, 296 of 128 ~58, 592 of 64 ~50 (the one-row GEMV: 42.8);
  // whole 32-row 7B step 0.425 -> 0.469 of HBM (profiles/r06_wide_step.txt)
  if (g_force_cfg == 1) return CFG(64, 64);
  if (g_force_cfg == 2) return CFG(64, 128);
  if (g_force_cfg == 3) return CFG(128, 128);
  if (M <= 32 && g_skinny64 && g_skinny32 && glds) return launch_cfg<32, 64, EPI, true>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
  if (M <= 64 && g_skinny64) return CFG(64, 64);
  if (t128 >= 200) return CFG(128, 128);
  // 64 x 128 tiles prefetch ONE K tile, the 64 x 64 kernel three: below ~3 tiles of 64 x 64 per CU and from K = 2048 the smaller
  // tile wins (Idefics2 down at T = 386: 183 -> 108 us, 7B down at T = 640: 241 -> 206, 7B qkv at T = 640: 38.8 -> 36.6; from
  // ~900 tiles the wider one is ahead again: profiles/r06_wide_step.txt, item 6)
  if (t64n >= 200 && !(g_midsize && glds && K >= 2048 && (long)vlm_cdiv(M, 64) * vlm_cdiv(N, 64) < 768)) return CFG(64, 128);
  return CFG(64, 64);
#undef CFG
}

}  // namespace

// 0 = automatic (LDS DMA when K % 64 == 0; 256x256 phased kernel for large shapes), 1 = always stage through
// registers (128x128 kernel), 2 = LDS DMA but never the 256x256 kernel, 3 = 256x256 kernel whenever it is legal.
// 4 = as 3 with the 2-phase variant of the 256x256 kernel, 5 = automatic with the 2-phase variant,
// 6 / 7 = as 3 with the tile width forced to 192 / 256 (3 picks it by last-round fill),
// 8 = LDS-DMA 128 kernel, never split-K; 9 = 128 kernel with split-K x4 forced (modes 1 / 2 also disable split-K).
// Test / A-B knob only.
extern "C" int vlm_gemm_set_staging(int mode) {
  g_force_cfg = 0;
  if (mode >= 100 && mode < 200) {       // tuning hook: plain kernels only, tile cfg = mode % 10, forced split count = (mode - 100) / 10 (0: never)
    g_force_regstage = false;
    g_tile256 = -1;
    g_force_cfg = mode % 10;
    const int sp = (mode - 100) / 10;
    g_splitk = sp > 1 ? sp : -1;
    return VLM_OK;
  }
  g_force_regstage = (mode == 1);
  g_tile256 = (mode == 3 || mode == 4 || mode == 6 || mode == 7 || mode == 10) ? 1 : (mode == 1 || mode == 2 || mode == 8 || mode == 9) ? -1 : 0;
  g_splitk = (mode == 1 || mode == 2 || mode == 8) ? -1 : mode == 9 ? 4 : 0;   // 8: no split-K, 9: split-K x4 forced
  vlm_gemm256_set_nf(mode == 6 ? 3 : mode == 7 ? 4 : 0);
  vlm_gemm256_set_persist(mode == 10 ? 1 : 0);   // 10: 256 kernel forced, persistent tile loop
  vlm_gemm256_set_variant((mode == 4 || mode == 5) ? 1 : 0);
#ifdef VLM_GEMM_ABLATION
  if (mode >= 11 && mode <= 13) {   // ablation probes (wrong results; see gemm256_bf16.hip)
    vlm_gemm256_set_variant(mode);
    g_tile256 = 1;
  }
#endif
  return VLM_OK;
}

static int gemm_dispatch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K,
                         int lda, int ldw, int ldc, int ldres, int epilogue, void* stream);

extern "C" int vlm_gemm_bf16(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N,
                             int K, int lda, int ldw, int ldc, int ldres, int epilogue, void* stream) {
  if (epilogue & ~(VLM_EPI_BIAS | VLM_EPI_GELU_FAST | VLM_EPI_GELU_ERF | VLM_EPI_RESIDUAL | VLM_EPI_SWIGLU)) return VLM_ERR_ARG;
  return gemm_dispatch(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, epilogue, stream);
}

extern "C" int vlm_gemm_bf16_rope2d(const void* A, const void* W, const void* bias, const void* cos_sin, void* C, int M,
                                    int N, int K, int lda, int ldw, int ldc, int head_dim, int rope_cols, void* stream) {
  if (!cos_sin || !bias || head_dim <= 0 || head_dim > 4095 || head_dim % 16 != 0 || rope_cols < 0 || rope_cols > N ||
      rope_cols % head_dim != 0)
    return VLM_ERR_ARG;
  return gemm_dispatch(A, W, bias, cos_sin, C, M, N, K, lda, ldw, ldc, head_dim | (rope_cols << 12),
                       VLM_EPI_BIAS | VLM_EPI_ROPE2D, stream);
}

static int gemm_dispatch(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K,
                         int lda, int ldw, int ldc, int ldres, int epilogue, void* stream) {
  if (!A || !W || !C || M < 0 || N <= 0 || K <= 0) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_BIAS) && !bias) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_RESIDUAL) && !res) return VLM_ERR_ARG;
  if (K % 8 != 0 || N % 8 != 0 || lda % 8 != 0 || ldw % 8 != 0 || ldc % 8 != 0) return VLM_ERR_SHAPE;
  if ((epilogue & VLM_EPI_RESIDUAL) && ldres % 8 != 0) return VLM_ERR_SHAPE;
  if ((epilogue & VLM_EPI_SWIGLU) && N % 16 != 0) return VLM_ERR_SHAPE;
  if (M == 0) return VLM_OK;
  hipStream_t st = (hipStream_t)stream;
  if (A && W && C && M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 &&
      (!(epilogue & VLM_EPI_RESIDUAL) || (res && ldres % 8 == 0)) && (!(epilogue & VLM_EPI_BIAS) || bias) && g_tile256 >= 0) {
    // measured on MI355X (scripts/gemm_bench.py 2 3): the phased 256x256 kernel wins from ~120 tiles up (140 tiles:
    // 566 vs 432 TF; 180: 838 vs 772; 540: 869 vs 828; 720: 1073-1087 vs 895-921 TF) and loses below (60 tiles:
    // 326-421 vs 471-530 TF), where the 128x128 kernel spreads the work over more CUs.  (g_tile256 == 1: test hook.)
    const long t256 = (long)vlm_cdiv(M, 256) * vlm_cdiv(N, 256);
    const bool fills = t256 >= 120 && !(g_skinny64 && M <= 64);      // (<= 64 rows: 64 x 64 tiles, see launch_epi)
    if (g_tile256 == 1 || fills) {
      const int rc = vlm_gemm256_try(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, epilogue, stream);
      if (rc >= 0) return rc;
    }
  }
#define GO(E) return launch_epi<E>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st)
  switch (epilogue) {
    case VLM_EPI_NONE: GO(VLM_EPI_NONE);
    case VLM_EPI_BIAS: GO(VLM_EPI_BIAS);
    case VLM_EPI_BIAS | VLM_EPI_ROPE2D: GO(VLM_EPI_BIAS | VLM_EPI_ROPE2D);
    case VLM_EPI_BIAS | VLM_EPI_GELU_FAST: GO(VLM_EPI_BIAS | VLM_EPI_GELU_FAST);
    case VLM_EPI_BIAS | VLM_EPI_GELU_ERF: GO(VLM_EPI_BIAS | VLM_EPI_GELU_ERF);
    case VLM_EPI_BIAS | VLM_EPI_RESIDUAL: GO(VLM_EPI_BIAS | VLM_EPI_RESIDUAL);
    case VLM_EPI_RESIDUAL: GO(VLM_EPI_RESIDUAL);
    case VLM_EPI_SWIGLU: GO(VLM_EPI_SWIGLU);
    default: return VLM_ERR_ARG;
  }
#undef GO
}

extern "C" int vlm_gemm_w4(const void* A, const void* Wq, const void* Wsb, const void* bias, const void* res, void* C, int M,
                           int N, int K, int lda, int ldc, int ldres, int epilogue, void* stream) {
  if (!A || !Wq || !Wsb || !C || M < 0 || N <= 0 || K <= 0) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_BIAS) && !bias) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_RESIDUAL) && !res) return VLM_ERR_ARG;
  if (K % 64 != 0 || N % 8 != 0 || lda % 8 != 0 || ldc % 8 != 0) return VLM_ERR_SHAPE;
  if ((epilogue & VLM_EPI_RESIDUAL) && ldres % 8 != 0) return VLM_ERR_SHAPE;
  if ((epilogue & VLM_EPI_SWIGLU) && N % 16 != 0) return VLM_ERR_SHAPE;
  if (M == 0) return VLM_OK;
  hipStream_t st = (hipStream_t)stream;
#define GO(E) return launch_epi_w4<E>(A, Wq, Wsb, bias, res, C, M, N, K, lda, ldc, ldres, st)
  switch (epilogue) {
    case VLM_EPI_NONE: GO(VLM_EPI_NONE);
    case VLM_EPI_BIAS: GO(VLM_EPI_BIAS);
    case VLM_EPI_BIAS | VLM_EPI_RESIDUAL: GO(VLM_EPI_BIAS | VLM_EPI_RESIDUAL);
    case VLM_EPI_RESIDUAL: GO(VLM_EPI_RESIDUAL);
    case VLM_EPI_SWIGLU: GO(VLM_EPI_SWIGLU);
    default: return VLM_ERR_ARG;
  }
#undef GO
}

// vlm_gemm_bf16 / vlm_gemm_w4 with a TAIL: when the GEMM takes the split-K route, its reduce launch also does what the caller
// would launch next on the reduced rows (VlmGemmTail, internal.h) and *tail_done = 1; otherwise the GEMM runs as usual,
// *tail_done = 0 and the caller launches the follower itself.
VLM_INTERNAL int vlm_gemm_bf16_tail(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K,
                                    int lda, int ldw, int ldc, int ldres, int epilogue, const VlmGemmTail* tail, int* tail_done,
                                    void* stream) {
  if (epilogue & ~(VLM_EPI_BIAS | VLM_EPI_RESIDUAL)) return VLM_ERR_ARG;
  t_tail = tail;
  t_tail_done = false;
  const int rc = gemm_dispatch(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, epilogue, stream);
  t_tail = nullptr;
  if (tail_done) *tail_done = t_tail_done ? 1 : 0;
  return rc;
}

VLM_INTERNAL int vlm_gemm_w4_tail(const void* A, const void* Wq, const void* Wsb, const void* bias, const void* res, void* C, int M,
                                  int N, int K, int lda, int ldc, int ldres, int epilogue, const VlmGemmTail* tail, int* tail_done,
                                  void* stream) {
  if (epilogue & ~(VLM_EPI_BIAS | VLM_EPI_RESIDUAL)) return VLM_ERR_ARG;
  t_tail = tail;
  t_tail_done = false;
  const int rc = vlm_gemm_w4(A, Wq, Wsb, bias, res, C, M, N, K, lda, ldc, ldres, epilogue, stream);
  t_tail = nullptr;
  if (tail_done) *tail_done = t_tail_done ? 1 : 0;
  return rc;
}

