// Long-K form of the skinny-M decode GEMM (gemv_mfma.hip) for gfx950: few 16-row tiles x a long contraction - the down
// projection of a batched decode step (Qwen2-VL-2B: N = 1536, K = 8960; 7B: 3584 x 18944; Mistral-7B: 4096 x 14336;
// reference call site mlx_vlm/models/qwen2_vl/language.py:123-133 at 3..16 rows, generate/ar.py:2584-2887).
//
// Why a form of its own (profiles/r03_batch16_kernel_stats_b.txt, r04_mfma2_shapes.txt): with 96..256 row tiles the
// general kernel has to split K over WORKGROUPS to fill the chip - 6 segments x 96 tiles at 2B widths - and every tile then
// ends in a cross-workgroup hand-off (fp32 partial tiles through memory, an arrival ticket, the last arriver's merge):
// 14.5 us for the 27.5 MB the one-row GEMV streams in 6.1.  Here ONE workgroup owns a tile's whole K: its 16 waves
// interleave the 128-wide chunks (wave w takes chunks w, w + 16, ...), two register sets per wave (the next chunk's loads
// are out before this one is multiplied), the 16 partial tiles meet in LDS in a fixed order, and nothing is handed from
// one workgroup to another.  96 workgroups then stream 286 KB each: a CU sustains that (16 waves x 2 x 8 KiB in flight),
// and the launch is one memory round trip + the stream instead of stream + hand-off.
// Weights: coalesced loads (one instruction = 4 rows x 256 contiguous bytes) transposed through a wave-private LDS region
// into A fragments, as in gemv_mfma.hip; the activations' B fragments come straight from global memory (x is 16 rows x K,
// L2 resident; every wave reads only its own chunks of it).  fp32 accumulation; epilogues: none / bias / residual / bias +
// residual with the rounding points of gemv_mfma.hip.  The summation order (chunks of a wave in sequence, then the waves
// 0..15) differs from the general kernel's: same fp32 bounds, tolerance-level agreement (tests/test_ops_gpu.py).
#include <stdlib.h>

#include "common.hpp"
#include "internal.h"
#include "../../include/vlm_hip.h"

namespace {

constexpr int NWV = 16;             // waves per workgroup
constexpr int WREG = 16 * 272;      // bytes of a wave's private transposition region (row pitch 272 B: conflict-free ds_read_b128)

struct LongKArgs {
  const bf16_t *x, *W, *bias, *res;
  bf16_t* y;
  int M, N, K, ldx, ldw, ldy, ldres, nblk;
};

template <int EPI, bool XLDS>
__global__ __launch_bounds__(NWV * 64) void gemv_mfma_longk_kernel(const LongKArgs a) {
  __shared__ __attribute__((aligned(16))) char wreg_all[(XLDS ? 2 : 1) * NWV * WREG];
  __shared__ float part[NWV * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r16 = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  char* wreg = wreg_all + wave * WREG;
  // instruction j of a chunk covers rows 4 j + g of the tile, 16 bytes at k offset 8 r16: 256 contiguous bytes per row
  const bf16_t* wr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wr[j] = a.W + (size_t)min(tile * 16 + 4 * j + g, a.N - 1) * a.ldw + r16 * 8;
  // the activations' chunk travels the same way (XLDS: 4 instructions of 4 rows x 256 contiguous bytes, transposed through a
  // second private region into B fragments: 8 cache lines per instruction) or as fragment-shaped loads straight from global
  // memory (batch row r16, k offset 32 s + 8 g: 16 half lines per instruction - measured slower, profiles/r04_longk_shapes.txt)
  const bf16_t* xr = XLDS ? a.x + (size_t)min(g, a.M - 1) * a.ldx + r16 * 8 : a.x + (size_t)min(r16, a.M - 1) * a.ldx + g * 8;
  char* xreg = wreg_all + (NWV + wave) * WREG;
  const int n_max = (a.nblk + NWV - 1) / NWV;           // rounds (uniform); this wave's chunk of round i: wave + 16 i
  auto load = [&](int i, u32x4_t (&w)[4], u32x4_t (&xf)[4]) __attribute__((always_inline)) {
    const size_t b = (size_t)min(wave + NWV * i, a.nblk - 1);          // surplus slot: a harmless re-read (L2), never multiplied
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wr[j] + b * 128));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // (XLDS: rows past M alias row M - 1: clamped per instruction)
      const bf16_t* p = XLDS ? a.x + (size_t)min(4 * s + g, a.M - 1) * a.ldx + r16 * 8 : xr + 32 * s;
      xf[s] = *reinterpret_cast<const u32x4_t*>(p + b * 128);
    }
  };
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  auto mul = [&](int i, const u32x4_t (&w)[4], const u32x4_t (&xf)[4]) __attribute__((always_inline)) {
    if (wave + NWV * i >= a.nblk) return;                               // (wave-uniform)
    // chunk -> the wave's region [16 rows][272 B] (as loaded: row 4 j + g, byte 16 r16), then the fragments; same-wave LDS
    // operations execute in order: no barrier
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4_t*>(wreg + (4 * j + g) * 272 + r16 * 16) = w[j];
    if (XLDS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4_t*>(xreg + (4 * j + g) * 272 + r16 * 16) = xf[j];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32x4_t af = *reinterpret_cast<const u32x4_t*>(wreg + r16 * 272 + s * 64 + g * 16);
      const u32x4_t bfr = XLDS ? *reinterpret_cast<const u32x4_t*>(xreg + r16 * 272 + s * 64 + g * 16) : xf[s];
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, bfr), acc, 0, 0, 0);
    }
  };
  u32x4_t wA[4], xA[4], wB[4], xB[4];
  load(0, wA, xA);
  for (int i = 0; i < n_max; i += 2) {
    load(i + 1, wB, xB);
    mul(i, wA, xA);
    load(i + 2, wA, xA);
    mul(i + 1, wB, xB);
  }
  // D[n = 4 g + i][m = r16] -> part[wave][n * 16 + m]; the 16 partial tiles are summed in the fixed order wave 0 .. 15
#pragma unroll
  for (int i = 0; i < 4; ++i) part[wave * 256 + (4 * g + i) * 16 + r16] = acc[i];
  __syncthreads();
  if (tid >= 256) return;
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) v += part[w * 256 + tid];
  const int n = tile * 16 + (tid >> 4), m = tid & 15;
  if (m < a.M && n < a.N) {
    if (EPI & VLM_EPI_BIAS) v += bf2f(a.bias[n]);
    if (EPI & VLM_EPI_RESIDUAL) v = rbf(v) + bf2f(a.res[(size_t)m * a.ldres + n]);
    a.y[(size_t)m * a.ldy + n] = f2bf(v);
  }
}

template <int EPI>
int launch_longk(const LongKArgs& a, hipStream_t st) {
  static const bool xlds = [] { const char* e = getenv("VLM_GEMV_MFMA_LONGK_XLDS"); return !e || atoi(e) != 0; }();   // A/B knob
  if (xlds) hipLaunchKernelGGL((gemv_mfma_longk_kernel<EPI, true>), dim3(vlm_cdiv(a.N, 16)), dim3(NWV * 64), 0, st, a);
  else hipLaunchKernelGGL((gemv_mfma_longk_kernel<EPI, false>), dim3(vlm_cdiv(a.N, 16)), dim3(NWV * 64), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

}  // namespace

// -> 0 done, > 0 error, -1 not this form's shape.  Taken for projections without a norm prologue whose K is long and whose
// row tiles alone cannot fill the chip (measured policy: VLM_GEMV_MFMA_LONGK = minimum K, 0 = off; profiles/r04_longk_shapes.txt)
VLM_INTERNAL int vlm_gemv_mfma_longk_try(const void* x, const void* W, const void* bias, const void* res, void* y, int M, int N,
                                         int K, int ldx, int ldw, int ldy, int ldres, int epilogue, void* stream) {
  static const int min_k = [] { const char* e = getenv("VLM_GEMV_MFMA_LONGK"); return e ? atoi(e) : 4096; }();
  if (min_k <= 0 || K < min_k || K % 128 || ldx % 8 || ldw % 8 || M < 1 || M > 16) return -1;
  const int n_tiles = vlm_cdiv(N, 16);
  static const int min_tiles = [] { const char* e = getenv("VLM_GEMV_MFMA_LONGK_MIN_TILES"); return e ? atoi(e) : 192; }();
  // fewer tiles than ~3/4 of the CUs: one workgroup per tile leaves the rest of the chip idle and a CU cannot pull a tile's
  // bytes fast enough to make up for it (2B down, 96 tiles: 20.8 vs 13.7 us); many tiles: the general kernel needs no K split
  if (n_tiles < min_tiles || n_tiles > 512) return -1;
  LongKArgs a{(const bf16_t*)x, (const bf16_t*)W, (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)y, M, N, K, ldx, ldw, ldy, ldres,
              K / 128};
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case VLM_EPI_NONE: return launch_longk<VLM_EPI_NONE>(a, st);
    case VLM_EPI_BIAS: return launch_longk<VLM_EPI_BIAS>(a, st);
    case VLM_EPI_RESIDUAL: return launch_longk<VLM_EPI_RESIDUAL>(a, st);
    case VLM_EPI_BIAS | VLM_EPI_RESIDUAL: return launch_longk<VLM_EPI_BIAS | VLM_EPI_RESIDUAL>(a, st);
    default: return -1;
  }
}
