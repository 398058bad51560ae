// Skinny-M decode GEMM, ROW-SLICE form (round 6): y[m][n] = epilogue( sum_k x[m][k] * W[n][k] ) for 3 <= M <= 16 batch rows when the
// projection has FEW output rows and a LONG K - the down projection of a batched decode step (reference mlp.py:6-14 at L = 1 under
// generate/ar.py:2584-2887; Qwen2-VL-2B: N = 1536, K = 8960).
//
// Why a third form.  gemv_mfma.hip / gemv_mfma2.hip cut the output into 16-row MFMA tiles: N = 1536 gives 96 tiles, and a compute
// unit pulls ~25 GB/s from HBM whatever its waves do (MI355X_MICROARCH.md ldsdma-fill / "~10 B/cyc/CU"), so 96 workgroups cannot
// stream 27.5 MB in time - the tiles' K is split over 5-6 workgroups whose fp32 partial tiles meet through memory (tickets, a
// last-arriver merge): 13.7-14.5 us for a launch whose one-row sibling takes 6.0 (profiles/r04_batch16_kernel_stats.txt).  The
// matrix pipe is idle in these launches (one MFMA per KiB of weights), so the tile need not be FULL: here a workgroup owns
// R = N / 256 output rows (6 at N = 1536) and their WHOLE K - the per-CU slice of the one-row GEMV - and multiplies them as the first R
// rows of a 16-row MFMA tile (rows R..15 of the accumulator are garbage of their own rows only and are dropped).  256 workgroups =
// one per CU, every byte of W requested within the first microseconds, no K split across workgroups, no workspace, no tickets; the
// 8 waves of a workgroup interleave the 128-wide K chunks and their partial tiles meet in LDS.
//
// Activations: every workgroup needs ALL of x (M x K: 286 KB at K = 8960), i.e. x is re-read 256 times through L2.  As MFMA B
// fragments (lane (m, g): 8 consecutive k of batch row m) a row-major x costs 16 half cache lines per wave instruction - the TA,
// not HBM, then paces the wave (profiles/r04_longk_shapes.txt).  So this form takes x TILED: xt[K / 8][16][8] bf16 (k group, batch
// row, 8 consecutive k): a fragment load is 64 lanes x 16 B = 1 KiB contiguous.  The producer writes that layout directly
// (gemv_mfma.hip's SwiGLU epilogue with MfmaArgs.y_tiled: the gate/up launch in front of the down projection).
#include <stdlib.h>

#include "common.hpp"
#include "internal.h"
#include "../../include/vlm_hip.h"

namespace {

constexpr int NW = 8;               // waves per workgroup
// SLOTS: chunks of a wave in flight (x fragments + weights: 16 + 4 NJ VGPRs each)
constexpr int WPITCH = 272;         // row pitch of a wave's private transposition region: conflict-free ds_read_b128 fragments
constexpr int WREGB = 16 * WPITCH;

struct RowsArgs {
  const bf16_t *xt, *W, *bias, *res;
  bf16_t* y;
  int M, N, K, ldw, ldy, ldres, R, nchunk;
  int dbg;      // timing probes (VLM_GEMV_ROWS_DBG, wrong results): 1 = every x fragment from chunk 0, 2 = every weight chunk from chunk 0
};

template <int EPI, int NJ, int SLOTS>          // NJ = weight load instructions per chunk = ceil(R / 4)
__global__ __launch_bounds__(64 * NW, 2) void gemv_mfma_rows_kernel(const RowsArgs a) {
  __shared__ __attribute__((aligned(16))) char s_wreg[NW * WREGB];
  __shared__ float part[NW * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, g = lane >> 4;
  char* wreg = s_wreg + wave * WREGB;
  const int row0 = (int)blockIdx.x * a.R;
  const int n_mine = (a.nchunk - wave + NW - 1) / NW;          // this wave's chunks: c = wave + NW i

  u32x4_t xf[SLOTS][4], wv[SLOTS][NJ];
  auto load = [&](int i, u32x4_t (&xs)[4], u32x4_t (&ws)[NJ]) __attribute__((always_inline)) {
    const int c = min(wave + NW * i, a.nchunk - 1);
    const int cx = a.dbg == 1 ? 0 : c, cw = a.dbg == 2 ? 0 : c;
    // x^T fragments of the chunk: step s = k 32 s + 8 g of batch row r16 -> k group 16 c + 4 s + g: 1 KiB contiguous per instruction
#pragma unroll
    for (int s = 0; s < 4; ++s)
      xs[s] = *reinterpret_cast<const u32x4_t*>(a.xt + ((size_t)(16 * cx + 4 * s + g) * 16 + r16) * 8);
    // weights: instruction j = rows 4 j + g of the slice, 16 bytes at k offset 8 r16 of the row's 256-byte chunk (streamed once)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int row = min(row0 + min(4 * j + g, a.R - 1), a.N - 1);
      ws[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(a.W + (size_t)row * a.ldw + (size_t)cw * 128 + r16 * 8));
    }
  };
#pragma unroll
  for (int i = 0; i < SLOTS; ++i)
    if (i < n_mine) load(i, xf[i], wv[i]);

  // epilogue operands of the thread that stores element (n_l = tid >> 4, m = tid & 15), requested behind the first slots
  const int n_l = (tid >> 4) & 15, m_e = tid & 15, n_e = row0 + n_l;
  const bool mine = tid < 256 && n_l < a.R && n_e < a.N && m_e < a.M;
  bf16_t e_b = 0, e_r = 0;
  if (mine) {
    if (EPI & VLM_EPI_BIAS) e_b = a.bias[n_e];
    if (EPI & VLM_EPI_RESIDUAL) e_r = a.res[(size_t)m_e * a.ldres + n_e];
  }

  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  auto consume = [&](u32x4_t (&xs)[4], u32x4_t (&ws)[NJ]) __attribute__((always_inline)) {
    // registers -> the wave's private region (same-wave LDS operations execute in order: no barrier), fragments, MFMAs
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<u32x4_t*>(wreg + (4 * j + g) * WPITCH + r16 * 16) = ws[j];
    // COMPILER TRAP (hipcc / ROCm 7.2, found in the .s): the fragment reads below take data OTHER lanes of the wave wrote; for the
    // optimiser a thread's own store never overlaps its own loads at kb = 2, 3 when NJ <= 2 (256 (g - r16) + 1088 j = 64 kb + ...
    // has no solution), so it hoisted those two ds_reads OUT of the chunk loop - every chunk then multiplied the first chunk's
    // (or stale) bytes.  The hardware executes a wave's LDS operations in order; the compiler needs to be told that memory changed.
    asm volatile("" ::: "memory");
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const u32x4_t af = *reinterpret_cast<const u32x4_t*>(wreg + r16 * WPITCH + kb * 64 + g * 16);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af), __builtin_bit_cast(bf16x8_t, xs[kb]), acc, 0, 0, 0);
    }
  };
  // slot i % SLOTS holds chunk i; it is refilled for chunk i + SLOTS the moment it has gone to LDS (the loop is unrolled by SLOTS so
  // that every slot index is a compile-time constant: runtime-indexed register arrays would live in scratch)
  for (int i0 = 0; i0 < n_mine; i0 += SLOTS) {
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      const int i = i0 + sl;
      if (i < n_mine) {                    // (wave-uniform)
        consume(xf[sl], wv[sl]);
        if (i + SLOTS < n_mine) load(i + SLOTS, xf[sl], wv[sl]);
      }
    }
  }

  // D[n = 4 g + q][m = r16] -> part[wave][n * 16 + m]; the 8 partial tiles are summed in a fixed order
#pragma unroll
  for (int q = 0; q < 4; ++q) part[wave * 256 + (4 * g + q) * 16 + r16] = acc[q];
  __syncthreads();
  if (mine) {
    float v = ((part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid])) +
              ((part[1024 + tid] + part[1280 + tid]) + (part[1536 + tid] + part[1792 + tid]));
    if (EPI & VLM_EPI_BIAS) v += bf2f(e_b);
    if (EPI & VLM_EPI_RESIDUAL) v = rbf(v) + bf2f(e_r);
    a.y[(size_t)m_e * a.ldy + n_e] = f2bf(v);
  }
}

template <int EPI>
int launch_rows(const RowsArgs& a, hipStream_t st) {
  const int grid = vlm_cdiv(a.N, a.R);
  static const int slots_env = [] { const char* e = getenv("VLM_GEMV_ROWS_SLOTS"); return e ? atoi(e) : 0; }();   // A/B knob
  if (a.R <= 4) hipLaunchKernelGGL((gemv_mfma_rows_kernel<EPI, 1, 4>), dim3(grid), dim3(64 * NW), 0, st, a);
  else if (a.R <= 8) {
    // (measured at 2B dims, 16 rows: 3 slots 9.7 us, 5: 10.0, 7: 10.0, 9 = everything up front: 10.6 - the launch is bound by the
    //  256 re-reads of x through L2, not by what a wave has in flight: profiles/r06_mfma_rows.txt)
    if (slots_env == 5) hipLaunchKernelGGL((gemv_mfma_rows_kernel<EPI, 2, 5>), dim3(grid), dim3(64 * NW), 0, st, a);
    else if (slots_env == 9) hipLaunchKernelGGL((gemv_mfma_rows_kernel<EPI, 2, 9>), dim3(grid), dim3(64 * NW), 0, st, a);
    else hipLaunchKernelGGL((gemv_mfma_rows_kernel<EPI, 2, 3>), dim3(grid), dim3(64 * NW), 0, st, a);
  } else if (a.R <= 12) hipLaunchKernelGGL((gemv_mfma_rows_kernel<EPI, 3, 3>), dim3(grid), dim3(64 * NW), 0, st, a);
  else hipLaunchKernelGGL((gemv_mfma_rows_kernel<EPI, 4, 3>), dim3(grid), dim3(64 * NW), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

}  // namespace

// -> 0 done, > 0 error, -1 shape not handled.  xt: the activations in the TILED layout [K / 8][16][8] bf16 (rows M..15 may hold
// anything finite or not: they only reach accumulator columns that are dropped).  Policy (measured, profiles/r06_mfma_rows.txt): taken
// for bf16 projections without a norm prologue, 3 <= M <= 16, K % 128 == 0, K >= 4096 and at most 16 rows per CU-sized slice.
VLM_INTERNAL int vlm_gemv_mfma_rows_ok(int M, int N, int K) {
  static const bool enabled = [] { const char* e = getenv("VLM_GEMV_MFMA_ROWS"); return !e || atoi(e) != 0; }();      // A/B knob
  return enabled && M >= 1 && M <= 16 && K % 128 == 0 && K >= 4096 && N >= 256 && vlm_cdiv(N, 256) <= 16;
}

VLM_INTERNAL int vlm_gemv_mfma_rows_try(const void* xt, const void* W, const void* bias, const void* res, void* y, int M, int N, int K,
                                        int ldw, int ldy, int ldres, int epilogue, void* stream) {
  if (!vlm_gemv_mfma_rows_ok(M, N, K) || ldw % 8 != 0) return -1;
  const int R = vlm_cdiv(N, 256);
  RowsArgs a{(const bf16_t*)xt, (const bf16_t*)W, (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)y, M, N, K, ldw, ldy, ldres, R, K / 128, 0};
  static const int dbg = [] { const char* e = getenv("VLM_GEMV_ROWS_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case VLM_EPI_NONE: return launch_rows<VLM_EPI_NONE>(a, st);
    case VLM_EPI_BIAS: return launch_rows<VLM_EPI_BIAS>(a, st);
    case VLM_EPI_RESIDUAL: return launch_rows<VLM_EPI_RESIDUAL>(a, st);
    case VLM_EPI_BIAS | VLM_EPI_RESIDUAL: return launch_rows<VLM_EPI_BIAS | VLM_EPI_RESIDUAL>(a, st);
    default: return -1;
  }
}
