// 256x256x64 bf16 MFMA GEMM for gfx950, software-pipelined at workgroup level ("phased" schedule).
//
// Same contract as gemm_bf16.hip (C[M,N] = epilogue(A[M,K] . W[N,K]^T), same fragments, same accumulation order
// per output element -> BIT-IDENTICAL results), used by vlm_gemm_bf16 for the large prefill GEMMs
// (reference nn.Linear call sites: mlx_vlm/models/qwen2_vl/vision.py:129-130,168-173; language.py:52-55,120;
// mlp.py:9-14).  What changes is the schedule - the 128x128 kernel stalls at every K tile (ds_read -> MFMA in
// every wave at once, one barrier, repeat); here:
//   * 8 waves (2 along M x 4 along N), each a 128x64 output block = 8x4 fragments of 16x16 (128 accumulator regs);
//   * a K tile is consumed in 4 PHASES of 16 MFMAs (one 32-row quarter of the wave's block x all 4 column
//     fragments x K = 64); the W fragments of a K tile are read from LDS once (phase 1), an A quarter per phase;
//   * the two wave rows run HALF A PHASE APART (the M-row-1 waves execute one extra barrier up front): on every
//     SIMD one wave issues ds_reads while the other one issues MFMAs, so the LDS pipe and the matrix pipe are both
//     busy all the time (waves w and w+4 share a SIMD);
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip), two K tiles resident (128 KiB),
//     every region refilled exactly two phases after its last read, i.e. 6 phases (~1.5 K tiles) before its next
//     use, ordered by COUNTED vmcnt waits (never 0 in the loop) one phase before the reading phase.
//
// Phase p of a wave:   ds_read(p) ; DMA issue(p) ; s_waitcnt vmcnt(N_p) ; BARRIER ; lgkmcnt(0) ; 16 MFMA ; BARRIER
// Barriers are numbered globally b = 0, 1, ...; wave row 0 runs phase p between barriers (2p-1, 2p+1), wave row 1
// between (2p, 2p+2).
//   RAW (DMA -> ds_read): a region read in phase p'' was waited for (every wave, its own pieces) in phase p''-1,
//     i.e. before barrier 2p''-2 / 2p''-1 <= the barrier that precedes the first read of phase p''.
//   WAR (ds_read -> DMA): reads of phase p are complete (lgkmcnt(0)) before barrier 2p+1 (row 0) / 2p+2 (row 1);
//     the refill is issued in phase p+2, after barrier 2p+3 / 2p+4.
// DMA issue list (T = current K tile, tile index clamped at the end so the counts never change):
//   phase 1: A rows [64,96) of T+1     phase 2: A rows [96,128) of T+1
//   phase 3: W (all) and A rows [0,32) of T+2     phase 4: A rows [32,64) of T+2
// = 1, 1, 5, 1 wave-instructions of 1 KiB; the waits are vmcnt(9), (9), (13), (9) (derivation in DESIGN.md).
#include <stdlib.h>

#include "common.hpp"
#include "../../include/vlm_hip.h"

#ifdef GEMM_STAMPS
// measurement build only (scripts/r05_gemm_stamps.py): wall-clock stamps (100 MHz s_memrealtime) of every workgroup of the
// 4-phase kernel - entry, first K tile landed, K loop done, epilogue converted, end - and the CU it ran on
__device__ unsigned long long g_gemm_stamps[16384][8];
extern "C" int vlm_debug_gemm_stamps(void* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * 8 * (size_t)n);
}
#define GST(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_gemm_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GST(i)
#endif

#ifndef GEMM_DMA_AUX
#define GEMM_DMA_AUX 0      // cache-policy bits of the LDS-DMA loads (measurement builds: -DGEMM_DMA_AUX=1 sc0, 2 nt, 16 sc1, ...)
#endif

namespace {

constexpr int TB = 256;                 // tile edge (M and N)
constexpr int BK = 64;
constexpr int ROWB = BK * 2;            // bytes per LDS row
constexpr int HALF = 128 * ROWB;        // one 128-row half tile: 16 KiB
constexpr int STAGE = 4 * HALF;         // A0 A1 W0 W1: 64 KiB per K tile
constexpr int C_LD = TB + 8;            // padded bf16 row of the epilogue tile
constexpr int LDS_BYTES = TB * C_LD * 2;   // 135168 >= 2 * STAGE
constexpr int GROUP_M = 4;              // tile rows per group of the in-XCD tile order

__device__ __forceinline__ int lds_off(int row, int slot) { return row * ROWB + ((slot ^ ((row >> 1) & 7)) << 4); }

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BARRIER()                       \
  do {                                  \
    __builtin_amdgcn_sched_barrier(0);  \
    __builtin_amdgcn_s_barrier();       \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)

template <int EPI, int NF = 4>
__device__ __forceinline__ void epilogue256(f32x4_t (&acc)[8][NF], char* smem, const bf16_t* __restrict__ bias,
                                            const bf16_t* __restrict__ res, bf16_t* __restrict__ C, int M, int N, int ldc,
                                            int ldres, int m0, int n0, int wr, int wc, int fr, int fs, int tid) {
  // ---- epilogue (as gemm_bf16.hip): lane holds D^T[n = nb + fs*4 + r][m = mb + fr]; bias / activation in registers
  //      with the reference's rounding points, tile transposed through LDS, coalesced 16-byte stores (+ residual).
  //      In TWO halves (round 5): fragment rows mi = 0..3 of every wave are converted and written first; while mi = 4..7
  //      are converted (the GELU epilogue is ~1770 VALU per lane and tile, 12 % of fc1) the 16-byte stores of the first
  //      half's rows are issued between the fragments - the store issue of one half runs under the arithmetic of the
  //      other instead of behind it.  Same values, same addresses.
  constexpr int TN = 64 * NF;          // tile width: 256 or 192 columns
  constexpr int C_LDN = TN + 8;
  constexpr bool SWI = (EPI & VLM_EPI_SWIGLU) != 0;
  constexpr int CPR = (SWI ? TN / 2 : TN) / 8;   // 16-byte chunks per output tile row
  constexpr int PER_HALF = 128 * CPR / 512;      // store chunks of one half per thread: 8 / 6 / 4 / 3
  static_assert(128 * CPR % 512 == 0, "a half tile's chunks must divide over the 512 threads");
  bf16_t* cs = reinterpret_cast<bf16_t*>(smem);
  const int n_out = SWI ? (N >> 1) : N, n0o = SWI ? (n0 >> 1) : n0;
  auto convert = [&](int mi) {
    const int ml = wr * 128 + mi * 16 + fr;
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) {
      const int nl = wc * 16 * NF + ni * 16 + fs * 4;
      const int n = min(n0 + nl, N - 4);
      float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      if (EPI & VLM_EPI_BIAS) {
        const uint2 b = *reinterpret_cast<const uint2*>(bias + n);
        v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
      }
      if (EPI & VLM_EPI_SWIGLU) {
        // interleaved (gate, up) rows of W -> N/2 outputs (reference mlp.py:9-14, activations.py:7-9)
        const float o0 = swiglu_(rbf(v[0]), rbf(v[1])), o1 = swiglu_(rbf(v[2]), rbf(v[3]));
        *reinterpret_cast<uint32_t*>(cs + ml * C_LDN + (nl >> 1)) = pack_bf2(o0, o1);
        continue;
      }
      if (EPI & VLM_EPI_GELU_FAST) {
        const vlm_f32x2_t g0 = gelu_fast2_(rbf2(vlm_f32x2_t{v[0], v[1]})), g1 = gelu_fast2_(rbf2(vlm_f32x2_t{v[2], v[3]}));
        v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
      }
      if (EPI & VLM_EPI_GELU_ERF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_(rbf(v[r]));
      }
      uint2 o;
      o.x = pack_bf2(v[0], v[1]);
      o.y = pack_bf2(v[2], v[3]);
      *reinterpret_cast<uint2*>(cs + ml * C_LDN + nl) = o;
    }
  };
  // chunk j (0 .. PER_HALF-1) of this thread in half hf: rows [0, 64) + [128, 192) (hf = 0) or [64, 128) + [192, 256)
  auto store = [&](int hf, int j) {
    const int c = tid + 512 * j;
    const int hr = c / CPR, cc = c % CPR;                       // row inside the half (0 .. 127), chunk inside the row
    const int row = (hr >> 6) * 128 + hf * 64 + (hr & 63);
    const int m = m0 + row, n = n0o + cc * 8;
    if (m < M && n < n_out) {
      uint4 u = *reinterpret_cast<const uint4*>(cs + row * C_LDN + cc * 8);
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
  };
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) convert(mi);
  __syncthreads();
  GST(3);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    convert(4 + k);
#pragma unroll
    for (int j = PER_HALF * k / 4; j < PER_HALF * (k + 1) / 4; ++j) store(0, j);
  }
  __syncthreads();
  GST(6);
#pragma unroll
  for (int j = 0; j < PER_HALF; ++j) store(1, j);
}

// ABL (ablation builds, scripts/gemm_bench.py): 1 = no DMA in the K loop, 2 = no ds_reads in the K loop,
// 3 = no barriers in the K loop - WRONG results, timing probes only.
// NF = 16-column fragments per wave: 4 -> 256x256 tile, 3 -> 256x192 (used when it fills the last round of workgroups
// better: N = 3840 of the ViT qkv projection gives 36 x 20 = 720 tiles = 2.8 rounds instead of 540 = 2.1).  W pieces per
// wave and phase 3's issue count become NF and NF + 1, the counted waits NF + 5 / 2 NF + 5.
template <int EPI, int ABL = 0, int NF = 4>
__global__ __launch_bounds__(512) void gemm256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                      const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                      bf16_t* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                                      int ldc, int ldres, int tiles_n, int nwg, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GST(0);
#ifdef GEMM_STAMPS
  const unsigned long long mt0 = __builtin_amdgcn_s_memtime();   // shader cycles: slot 7 = cycles entry -> end (effective clock)
  if (threadIdx.x == 0 && blockIdx.x < 16384) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_gemm_stamps[blockIdx.x][5] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  // XCD-aware bijective remap: blocks with the same (bid % 8) share an L2.
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // grouped order inside the XCD's range: the ~32 tiles an XCD runs at once form a group_m x (32 / group_m) block
  // (4 A row-blocks + 8 W row-blocks through its L2 instead of 1 + 32): 8192^3 1160 -> 1409 TF
  int pid_m, pid_n;
  {
    const int tiles_m = nwg / tiles_n, per_group = group_m * tiles_n, gid = bid / per_group;
    const int first_m = gid * group_m, gsz = min(tiles_m - first_m, group_m), r = bid - gid * per_group;
    pid_m = first_m + r % gsz;
    pid_n = r / gsz;
  }
  constexpr int WH = 32 * NF;                   // rows of one W half tile (128 or 96)
  const int m0 = pid_m * TB, n0 = pid_n * 64 * NF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = uw >> 2, wc = uw & 3;          // wave row (M half) / wave column (16*NF-wide N quarter)
  const int nk = K / BK;

  // ---- DMA sources: lane l of a 1 KiB piece covers row (l >> 3), k-slot (l & 7) ^ swizzle(row)
  const int rl = lane >> 3, sp = lane & 7;
  const bf16_t* wsrc[NF];  // W pieces NF*uw .. NF*uw+NF-1: half (uw >> 2), rows 8*((uw & 3)*NF + j) .. +8
  const bf16_t* asrc[4];   // A quarter q: half (uw >> 2), rows 32*q + 8*(uw & 3) .. +8
  int wdst[NF], adst[4];   // LDS byte offsets inside a stage
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int prow = 8 * ((uw & 3) * NF + j);                // piece's first row inside its half
    const int row = prow + rl;
    wsrc[j] = W + (size_t)min(n0 + (uw >> 2) * WH + row, N - 1) * ldw + ((sp ^ ((row >> 1) & 7)) << 3);
    wdst[j] = 2 * HALF + (uw >> 2) * HALF + prow * ROWB;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int arow0 = 32 * j + 8 * (uw & 3);
    const int arow = arow0 + rl;
    asrc[j] = A + (size_t)min(m0 + (uw >> 2) * 128 + arow, M - 1) * lda + ((sp ^ ((arow >> 1) & 7)) << 3);
    adst[j] = (uw >> 2) * HALF + arow0 * ROWB;
  }
  auto dma = [&](const bf16_t* src, int kt, int lds_byte) {
    if (ABL == 1 && kt >= 2) return;
    const int ktc = min(kt, nk - 1);   // past the end: a harmless reload (keeps the vmcnt arithmetic uniform)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)ktc * BK),
                                     (__attribute__((address_space(3))) void*)(smem + lds_byte), 16, 0, GEMM_DMA_AUX);
  };
  auto issue_w = [&](int kt) {
    const int base = (kt & 1) * STAGE;
#pragma unroll
    for (int j = 0; j < NF; ++j) dma(wsrc[j], kt, base + wdst[j]);
  };
  auto issue_a = [&](int kt, int q) { dma(asrc[q], kt, (kt & 1) * STAGE + adst[q]); };

  f32x4_t acc[8][NF];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#define WAIT_A()              \
  do {                        \
    if (NF == 4) VMCNT(9);    \
    else VMCNT(8);            \
  } while (0)   /* NF + 5 */
#define WAIT_B()              \
  do {                        \
    if (NF == 4) VMCNT(13);   \
    else VMCNT(11);           \
  } while (0)   /* 2 NF + 5 */
  // ---- prologue: the issue order of the steady state (see header), then the wait of "phase 4 of tile -1"
  issue_w(0); issue_a(0, 0); issue_a(0, 1); issue_a(0, 2); issue_a(0, 3);
  issue_w(1); issue_a(1, 0); issue_a(1, 1);
  WAIT_A();
  BARRIER();
  GST(1);
  if (wr == 1) BARRIER();   // the second wave row runs half a phase behind

  bf16x8_t wf[NF][2], af[2][2];
  const int fr = lane & 15, fs = lane >> 4;
  auto read_w = [&](int kt) {
    if (ABL == 2 && kt >= 1) return;
    const char* ws = smem + (kt & 1) * STAGE + 2 * HALF + (wc >> 1) * HALF;
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wf[n][ks] = *reinterpret_cast<const bf16x8_t*>(ws + lds_off((wc & 1) * 16 * NF + n * 16 + fr, ks * 4 + fs));
  };
  auto read_a = [&](int kt, int q) {
    if (ABL == 2 && kt >= 1) return;
    const char* as = smem + (kt & 1) * STAGE + wr * HALF;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        af[m][ks] = *reinterpret_cast<const bf16x8_t*>(as + lds_off(q * 32 + m * 16 + fr, ks * 4 + fs));
  };

#define LOOPBAR()            \
  do {                       \
    if (ABL != 3) BARRIER(); \
  } while (0)
#define MFMA_PHASE(Q)                                                                                              \
  do {                                                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
      _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                                \
        _Pragma("unroll") for (int n = 0; n < NF; ++n)                                                             \
          acc[2 * (Q) + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n][ks], af[m][ks], acc[2 * (Q) + m][n], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
  } while (0)

  for (int kt = 0; kt < nk; ++kt) {
    // ---- phase 1
    read_w(kt);
    read_a(kt, 0);
    __builtin_amdgcn_sched_barrier(0);
    issue_a(kt + 1, 2);
    WAIT_A();
    LOOPBAR();
    LGKM0();
    MFMA_PHASE(0);
    LOOPBAR();
    // ---- phase 2
    read_a(kt, 1);
    __builtin_amdgcn_sched_barrier(0);
    issue_a(kt + 1, 3);
    WAIT_A();
    LOOPBAR();
    LGKM0();
    MFMA_PHASE(1);
    LOOPBAR();
    // ---- phase 3
    read_a(kt, 2);
    __builtin_amdgcn_sched_barrier(0);
    issue_w(kt + 2);
    issue_a(kt + 2, 0);
    WAIT_B();
    LOOPBAR();
    LGKM0();
    MFMA_PHASE(2);
    LOOPBAR();
    // ---- phase 4
    read_a(kt, 3);
    __builtin_amdgcn_sched_barrier(0);
    issue_a(kt + 2, 1);
    WAIT_A();
    LOOPBAR();
    LGKM0();
    MFMA_PHASE(3);
    LOOPBAR();
  }
#undef MFMA_PHASE
#undef LOOPBAR
#undef WAIT_A
#undef WAIT_B
  if (wr == 0) BARRIER();   // barrier counts must match across the workgroup
  VMCNT(0);                 // the clamped reloads past the last K tile still target LDS
  BARRIER();
  GST(2);

  epilogue256<EPI, NF>(acc, smem, bias, res, C, M, N, ldc, ldres, m0, n0, wr, wc, fr, fs, tid);
  GST(4);
#ifdef GEMM_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 16384) g_gemm_stamps[blockIdx.x][7] = __builtin_amdgcn_s_memtime() - mt0;
#endif
}

// ---- persistent form of the 4-phase kernel: min(tiles, 256) workgroups walk the tiles (tile = b, b + grid, ...: the same
// XCD every time).  After a tile's K loop the NEXT tile's first 14 LDS-DMA pieces are issued BEFORE the epilogue, which
// runs out of a separate 17 KiB LDS chunk (32 rows at a time, 2 barriers per chunk) - the ~2 us of cold DMA latency at
// the start of a tile and the bias/activation/store pass at its end overlap instead of adding up (K = 1280 shapes have
// only 20 K tiles per tile to amortise them over).
template <int EPI, int NF>
__global__ __launch_bounds__(512) void gemm256p_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                       const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                       bf16_t* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                                       int ldc, int ldres, int tiles_n, int nwg, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WH = 32 * NF, TN = 64 * NF, C_LDN = TN + 8;
  constexpr bool SWI = (EPI & VLM_EPI_SWIGLU) != 0;
  bf16_t* cs = reinterpret_cast<bf16_t*>(smem + 2 * STAGE);     // epilogue chunk: 32 rows x (TN + 8)
  const int tid = threadIdx.x, lane = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = uw >> 2, wc = uw & 3;
  const int nk = K / BK;
  const int rl = lane >> 3, sp = lane & 7, fr = lane & 15, fs = lane >> 4;
  const int tiles_m = nwg / tiles_n;

  auto coords = [&](int tile, int& m0, int& n0) {
    const int q = nwg >> 3, r = nwg & 7, xcd = tile & 7, idx = tile >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = group_m * tiles_n, gid = bid / per_group;
    const int first_m = gid * group_m, gsz = min(tiles_m - first_m, group_m), rr = bid - gid * per_group;
    m0 = (first_m + rr % gsz) * TB;
    n0 = (rr / gsz) * TN;
  };
  const bf16_t* wsrc[NF];
  const bf16_t* asrc[4];
  int wdst[NF], adst[4];
#pragma unroll
  for (int j = 0; j < NF; ++j) wdst[j] = 2 * HALF + (uw >> 2) * HALF + 8 * ((uw & 3) * NF + j) * ROWB;
#pragma unroll
  for (int j = 0; j < 4; ++j) adst[j] = (uw >> 2) * HALF + (32 * j + 8 * (uw & 3)) * ROWB;
  auto sources = [&](int m0, int n0) {
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int row = 8 * ((uw & 3) * NF + j) + rl;
      wsrc[j] = W + (size_t)min(n0 + (uw >> 2) * WH + row, N - 1) * ldw + ((sp ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int arow = 32 * j + 8 * (uw & 3) + rl;
      asrc[j] = A + (size_t)min(m0 + (uw >> 2) * 128 + arow, M - 1) * lda + ((sp ^ ((arow >> 1) & 7)) << 3);
    }
  };
  auto dma = [&](const bf16_t* src, int kt, int lds_byte) {
    const int ktc = min(kt, nk - 1);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)ktc * BK),
                                     (__attribute__((address_space(3))) void*)(smem + lds_byte), 16, 0, GEMM_DMA_AUX);
  };
  auto issue_w = [&](int kt) {
    const int base = (kt & 1) * STAGE;
#pragma unroll
    for (int j = 0; j < NF; ++j) dma(wsrc[j], kt, base + wdst[j]);
  };
  auto issue_a = [&](int kt, int q) { dma(asrc[q], kt, (kt & 1) * STAGE + adst[q]); };
  auto prologue = [&]() {
    issue_w(0); issue_a(0, 0); issue_a(0, 1); issue_a(0, 2); issue_a(0, 3);
    issue_w(1); issue_a(1, 0); issue_a(1, 1);
  };
#define WAIT_A()              \
  do {                        \
    if (NF == 4) VMCNT(9);    \
    else VMCNT(8);            \
  } while (0)
#define WAIT_B()              \
  do {                        \
    if (NF == 4) VMCNT(13);   \
    else VMCNT(11);           \
  } while (0)

  bf16x8_t wf[NF][2], af[2][2];
  auto read_w = [&](int kt) {
    const char* ws = smem + (kt & 1) * STAGE + 2 * HALF + (wc >> 1) * HALF;
#pragma unroll
    for (int n = 0; n < NF; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wf[n][ks] = *reinterpret_cast<const bf16x8_t*>(ws + lds_off((wc & 1) * 16 * NF + n * 16 + fr, ks * 4 + fs));
  };
  auto read_a = [&](int kt, int q) {
    const char* as = smem + (kt & 1) * STAGE + wr * HALF;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        af[m][ks] = *reinterpret_cast<const bf16x8_t*>(as + lds_off(q * 32 + m * 16 + fr, ks * 4 + fs));
  };

  int tile = blockIdx.x, m0, n0;
  coords(tile, m0, n0);
  sources(m0, n0);
  prologue();
  while (true) {
    WAIT_A();
    BARRIER();
    if (wr == 1) BARRIER();
    f32x4_t acc[8][NF];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#define MFMA_PHASE_P(Q)                                                                                            \
  do {                                                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
      _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                                \
        _Pragma("unroll") for (int n = 0; n < NF; ++n)                                                             \
          acc[2 * (Q) + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n][ks], af[m][ks], acc[2 * (Q) + m][n], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
  } while (0)
    for (int kt = 0; kt < nk; ++kt) {
      read_w(kt);
      read_a(kt, 0);
      __builtin_amdgcn_sched_barrier(0);
      issue_a(kt + 1, 2);
      WAIT_A();
      BARRIER();
      LGKM0();
      MFMA_PHASE_P(0);
      BARRIER();
      read_a(kt, 1);
      __builtin_amdgcn_sched_barrier(0);
      issue_a(kt + 1, 3);
      WAIT_A();
      BARRIER();
      LGKM0();
      MFMA_PHASE_P(1);
      BARRIER();
      read_a(kt, 2);
      __builtin_amdgcn_sched_barrier(0);
      issue_w(kt + 2);
      issue_a(kt + 2, 0);
      WAIT_B();
      BARRIER();
      LGKM0();
      MFMA_PHASE_P(2);
      BARRIER();
      read_a(kt, 3);
      __builtin_amdgcn_sched_barrier(0);
      issue_a(kt + 2, 1);
      WAIT_A();
      BARRIER();
      LGKM0();
      MFMA_PHASE_P(3);
      BARRIER();
    }
#undef MFMA_PHASE_P
    if (wr == 0) BARRIER();
    VMCNT(0);      // the clamped reloads past the last K tile must have LANDED before the next tile's pieces go to the
    BARRIER();     // same LDS addresses (two DMAs to one address may land in either order)

    const int next = tile + (int)gridDim.x;
    const bool has_next = next < nwg;
    int m0n = 0, n0n = 0;
    if (has_next) {
      coords(next, m0n, n0n);
      sources(m0n, n0n);
      prologue();           // flies under the epilogue below
    }
    // ---- epilogue in 8 chunks of 32 rows (m fragment mi of both wave rows) through the separate LDS chunk
    const int n_out = SWI ? (N >> 1) : N, n0o = SWI ? (n0 >> 1) : n0;
    constexpr int CPR = (SWI ? TN / 2 : TN) / 8;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int ml = wr * 16 + fr;
#pragma unroll
      for (int ni = 0; ni < NF; ++ni) {
        const int nl = wc * 16 * NF + ni * 16 + fs * 4;
        const int n = min(n0 + nl, N - 4);
        float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
        if (EPI & VLM_EPI_BIAS) {
          const uint2 b = *reinterpret_cast<const uint2*>(bias + n);
          v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
        }
        if (SWI) {
          const float o0 = swiglu_(rbf(v[0]), rbf(v[1])), o1 = swiglu_(rbf(v[2]), rbf(v[3]));
          *reinterpret_cast<uint32_t*>(cs + ml * C_LDN + (nl >> 1)) = pack_bf2(o0, o1);
          continue;
        }
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
        *reinterpret_cast<uint2*>(cs + ml * C_LDN + nl) = o;
      }
      LGKM0();      // raw barriers: __syncthreads() would also drain vmcnt(0), i.e. wait here for the next tile's DMA
      BARRIER();
      for (int c = tid; c < 32 * CPR; c += 512) {
        const int row = c / CPR, cc = c % CPR;
        const int m = m0 + (row >> 4) * 128 + mi * 16 + (row & 15), n = n0o + cc * 8;
        if (m < M && n < n_out) {
          uint4 u = *reinterpret_cast<const uint4*>(cs + row * C_LDN + cc * 8);
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
      BARRIER();    // the chunk's ds_reads have returned (their data went into the stores above)
    }
    if (!has_next) break;
    tile = next;
    m0 = m0n;
    n0 = n0n;
  }
#undef WAIT_A
#undef WAIT_B
}

// ---- variant B: the same tile and fragments, TWO phases of 32 MFMAs per K tile (half the barriers per MFMA).
// With only two phases a region cannot wait two phases for its refill and still arrive in time, so ALL DMA is issued
// by the wave row that runs half a phase behind (row 1): its issue point in phase p+1 lies after barrier 2p+2, by
// which every wave's reads of phase p are complete - a refill ONE phase after the last read is safe from there.
//   phase 1 (T): read W(T) + A rows [0,64);  row 1 issues A rows [64,128) of T+1 (4 x 1 KiB per wave)
//   phase 2 (T): read A rows [64,128);       row 1 issues W(T+2) and A rows [0,64) of T+2 (8 + 4 per wave)
// Every piece is waited for one phase before it is read; 16 newer pieces are outstanding then -> vmcnt(16) always.
template <int EPI>
__global__ __launch_bounds__(512) void gemm256b_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                       const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                       bf16_t* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                                       int ldc, int ldres, int tiles_n, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int pid_m, pid_n;
  {
    const int tiles_m = nwg / tiles_n, per_group = GROUP_M * tiles_n, gid = bid / per_group;
    const int first_m = gid * GROUP_M, gsz = min(tiles_m - first_m, GROUP_M), r = bid - gid * per_group;
    pid_m = first_m + r % gsz;
    pid_n = r / gsz;
  }
  const int m0 = pid_m * TB, n0 = pid_n * TB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int uw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = uw >> 2, wc = uw & 3;
  const int nk = K / BK;
  const int g = uw & 3;                          // DMA issuer index among the row-1 waves
  const int rl = lane >> 3, sp = lane & 7;

  // piece tables of issuer g: W pieces 8g..8g+7 (half g>>1, rows 8*((g&1)*8 + j)); A_lo / A_hi pieces 4g..4g+3
  // (half g>>1, rows 8*((g&1)*4 + j) (+64))
  const bf16_t* wsrc[8];
  const bf16_t* alsrc[4];
  const bf16_t* ahsrc[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = 8 * ((g & 1) * 8 + j) + rl;
    wsrc[j] = W + (size_t)min(n0 + (g >> 1) * 128 + row, N - 1) * ldw + ((sp ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * ((g & 1) * 4 + j) + rl;
    alsrc[j] = A + (size_t)min(m0 + (g >> 1) * 128 + row, M - 1) * lda + ((sp ^ ((row >> 1) & 7)) << 3);
    const int rowh = row + 64;
    ahsrc[j] = A + (size_t)min(m0 + (g >> 1) * 128 + rowh, M - 1) * lda + ((sp ^ ((rowh >> 1) & 7)) << 3);
  }
  auto dma = [&](const bf16_t* src, int kt, int lds_byte) {
    const int ktc = min(kt, nk - 1);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)ktc * BK),
                                     (__attribute__((address_space(3))) void*)(smem + lds_byte), 16, 0, GEMM_DMA_AUX);
  };
  auto issue_w = [&](int kt) {
    const int base = (kt & 1) * STAGE + 2 * HALF + (g >> 1) * HALF + (g & 1) * 8 * 8 * ROWB;
#pragma unroll
    for (int j = 0; j < 8; ++j) dma(wsrc[j], kt, base + j * 8 * ROWB);
  };
  auto issue_alo = [&](int kt) {
    const int base = (kt & 1) * STAGE + (g >> 1) * HALF + (g & 1) * 4 * 8 * ROWB;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma(alsrc[j], kt, base + j * 8 * ROWB);
  };
  auto issue_ahi = [&](int kt) {
    const int base = (kt & 1) * STAGE + (g >> 1) * HALF + 64 * ROWB + (g & 1) * 4 * 8 * ROWB;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma(ahsrc[j], kt, base + j * 8 * ROWB);
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (wr == 1) {   // prologue in steady-state issue order
    issue_w(0); issue_alo(0); issue_ahi(0);
    issue_w(1); issue_alo(1);
  }
  VMCNT(16);
  BARRIER();
  if (wr == 1) BARRIER();

  bf16x8_t wf[4][2], af[4][2];
  const int fr = lane & 15, fs = lane >> 4;
  auto read_w = [&](int kt) {
    const char* ws = smem + (kt & 1) * STAGE + 2 * HALF + (wc >> 1) * HALF;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        wf[n][ks] = *reinterpret_cast<const bf16x8_t*>(ws + lds_off((wc & 1) * 64 + n * 16 + fr, ks * 4 + fs));
  };
  auto read_a = [&](int kt, int hh) {
    const char* as = smem + (kt & 1) * STAGE + wr * HALF;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        af[m][ks] = *reinterpret_cast<const bf16x8_t*>(as + lds_off(hh * 64 + m * 16 + fr, ks * 4 + fs));
  };
#define MFMA_PHASE_B(HH)                                                                                           \
  do {                                                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                \
        _Pragma("unroll") for (int n = 0; n < 4; ++n)                                                              \
          acc[4 * (HH) + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n][ks], af[m][ks], acc[4 * (HH) + m][n], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
  } while (0)

  for (int kt = 0; kt < nk; ++kt) {
    // ---- phase 1
    read_w(kt);
    read_a(kt, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) issue_ahi(kt + 1);
    VMCNT(16);
    BARRIER();
    LGKM0();
    MFMA_PHASE_B(0);
    BARRIER();
    // ---- phase 2
    read_a(kt, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) {
      issue_w(kt + 2);
      issue_alo(kt + 2);
    }
    VMCNT(16);
    BARRIER();
    LGKM0();
    MFMA_PHASE_B(1);
    BARRIER();
  }
#undef MFMA_PHASE_B
  if (wr == 0) BARRIER();
  VMCNT(0);
  BARRIER();
  epilogue256<EPI>(acc, smem, bias, res, C, M, N, ldc, ldres, m0, n0, wr, wc, fr, fs, tid);
}

int g_variant = 0;   // 0 = 4 phases of 16 MFMAs per K tile, 1 = 2 phases of 32 (vlm_gemm256_set_variant, A/B knob)
int g_nf = 0;        // 0 = pick the tile width (256 / 192) by last-round fill, 3 / 4 = forced (tests)

int g_persist = 0;   // 1 = persistent tile loop with the next tile's DMA under the epilogue (A/B knob; default below)

template <int EPI, int NF>
int launch256_nf(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
                 int ldw, int ldc, int ldres, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI, 0, NF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return VLM_ERR_HIP + (int)e;
    attr_set = true;
  }
  const int tiles_m = vlm_cdiv(M, TB), tiles_n = vlm_cdiv(N, 64 * NF), nwg = tiles_m * tiles_n;
  if (g_persist == 1 && nwg > 256) {
    static bool attr_p = false;
    constexpr int LDS_P = 2 * STAGE + 32 * (64 * NF + 8) * 2;
    if (!attr_p) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<EPI, NF>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_P);
      if (e != hipSuccess) return VLM_ERR_HIP + (int)e;
      attr_p = true;
    }
    hipLaunchKernelGGL((gemm256p_kernel<EPI, NF>), dim3(256), dim3(512), LDS_P, st, (const bf16_t*)A, (const bf16_t*)W,
                       (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)C, M, N, K, lda, ldw, ldc, ldres, tiles_n, nwg,
                       GROUP_M);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
  }
  static const int group_m = getenv("VLM_GEMM_GROUP_M") ? atoi(getenv("VLM_GEMM_GROUP_M")) : GROUP_M;   // A/B knob
  hipLaunchKernelGGL((gemm256_kernel<EPI, 0, NF>), dim3(nwg), dim3(512), LDS_BYTES, st, (const bf16_t*)A, (const bf16_t*)W,
                     (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)C, M, N, K, lda, ldw, ldc, ldres, tiles_n, nwg,
                     group_m > 0 ? group_m : GROUP_M);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

// fraction of the machine doing useful work: occupancy of the workgroup rounds (256 CUs, one workgroup each) x the
// fraction of the tiles' columns that exist
inline double tile_score(int M, int N, int tn) {
  const long tiles = (long)vlm_cdiv(M, TB) * vlm_cdiv(N, tn), rounds = (tiles + 255) / 256;
  return (double)tiles / (double)(rounds * 256) * ((double)N / ((double)vlm_cdiv(N, tn) * tn));
}

template <int EPI>
int launch256(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
              int ldw, int ldc, int ldres, hipStream_t st) {
  if (g_variant == 1) {
    static bool attr_b = false;
    if (!attr_b) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256b_kernel<EPI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) return VLM_ERR_HIP + (int)e;
      attr_b = true;
    }
    const int tiles_m = vlm_cdiv(M, TB), tiles_n = vlm_cdiv(N, TB), nwg = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm256b_kernel<EPI>), dim3(nwg), dim3(512), LDS_BYTES, st, (const bf16_t*)A, (const bf16_t*)W,
                       (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)C, M, N, K, lda, ldw, ldc, ldres, tiles_n, nwg);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
  }
#ifdef VLM_GEMM_ABLATION   // timing probes with WRONG results: never in the shipped library (build with -DVLM_GEMM_ABLATION)
  if (g_variant >= 11 && g_variant <= 13 && EPI == VLM_EPI_NONE) {
    static bool abl_attr = false;
    if (!abl_attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<VLM_EPI_NONE, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<VLM_EPI_NONE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<VLM_EPI_NONE, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      abl_attr = true;
    }
    const int tiles_m = vlm_cdiv(M, TB), tiles_n = vlm_cdiv(N, TB), nwg = tiles_m * tiles_n;
#define ABL_GO(V)                                                                                                         \
  hipLaunchKernelGGL((gemm256_kernel<VLM_EPI_NONE, V>), dim3(nwg), dim3(512), LDS_BYTES, st, (const bf16_t*)A,            \
                     (const bf16_t*)W, (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)C, M, N, K, lda, ldw, ldc, ldres, \
                     tiles_n, nwg, GROUP_M)
    if (g_variant == 11) ABL_GO(1);
    else if (g_variant == 12) ABL_GO(2);
    else ABL_GO(3);
#undef ABL_GO
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
  }
#endif
  // 256x192 tiles when they keep more of the machine busy (ViT qkv: N = 3840 -> 720 tiles = 2.8 rounds instead of
  // 540 = 2.1; N = 1280 -> 252 tiles = one full round instead of 180); 3 % handicap for the smaller tile
  const bool nf3 = g_nf == 3 || (g_nf == 0 && 0.97 * tile_score(M, N, 192) > tile_score(M, N, 256));
  return nf3 ? launch256_nf<EPI, 3>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st)
             : launch256_nf<EPI, 4>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
}

}  // namespace

__attribute__((visibility("hidden"))) void vlm_gemm256_set_variant(int v) { g_variant = v; }
__attribute__((visibility("hidden"))) void vlm_gemm256_set_nf(int nf) { g_nf = nf; }
__attribute__((visibility("hidden"))) void vlm_gemm256_set_persist(int p) { g_persist = p; }

// Internal entry (C++ linkage, called by vlm_gemm_bf16's dispatcher).  Returns -1 when the shape / epilogue is not
// one this kernel takes (the caller then uses the 128x128 kernel).  Needs K % 64 == 0, K >= 128, N % 8 == 0.
__attribute__((visibility("hidden"))) int vlm_gemm256_try(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K,
                    int lda, int ldw, int ldc, int ldres, int epilogue, void* stream) {
  if (K % BK != 0 || K < 2 * BK || N < 8) return -1;
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case VLM_EPI_NONE: return launch256<VLM_EPI_NONE>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
    case VLM_EPI_BIAS: return launch256<VLM_EPI_BIAS>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
    case VLM_EPI_BIAS | VLM_EPI_ROPE2D:
      if (g_persist) return -1;          // the persistent variant has its own chunked epilogue (no rope form)
      return launch256<VLM_EPI_BIAS | VLM_EPI_ROPE2D>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
    case VLM_EPI_BIAS | VLM_EPI_GELU_FAST:
      return launch256<VLM_EPI_BIAS | VLM_EPI_GELU_FAST>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
    case VLM_EPI_BIAS | VLM_EPI_GELU_ERF:
      return launch256<VLM_EPI_BIAS | VLM_EPI_GELU_ERF>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
    case VLM_EPI_BIAS | VLM_EPI_RESIDUAL:
      return launch256<VLM_EPI_BIAS | VLM_EPI_RESIDUAL>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
    case VLM_EPI_RESIDUAL: return launch256<VLM_EPI_RESIDUAL>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
    case VLM_EPI_SWIGLU:
      if (N % 16 != 0) return -1;
      return launch256<VLM_EPI_SWIGLU>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
    default: return -1;
  }
}
