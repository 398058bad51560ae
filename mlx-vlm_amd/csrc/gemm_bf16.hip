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
#include "common.cuh"
#include "../../include/vlm_hip.h"

namespace {

constexpr int BK = 64;          // k elements per LDS tile
constexpr int ROWB = BK * 2;    // bytes per tile row (128)

__device__ __forceinline__ int lds_off(int row, int slot) { return row * ROWB + ((slot ^ ((row >> 1) & 7)) << 4); }

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                        const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                        bf16_t* __restrict__ C, int M, int N, int K, int lda, int ldw,
                                                        int ldc, int ldres, int tiles_n, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 16, NT = WN / 16;
  constexpr int A_PER = BM * 8 / 256, W_PER = BN * 8 / 256;
  constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;

  // XCD-aware bijective remap: blocks with the same (bid % 8) share an L2.
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
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
      rw[i] = (gk < K) ? *reinterpret_cast<const uint4*>(W + (size_t)gn * ldw + gk) : make_uint4(0, 0, 0, 0);
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
      *reinterpret_cast<uint4*>(ws + lds_off(c >> 3, c & 7)) = rw[i];
    }
  };

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) gload(kt + 1);
    const char* as = smem + (kt & 1) * STAGE;
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
    if (more) lstore((kt + 1) & 1);
    __syncthreads();
  }

  // epilogue: lane holds D^T[n = nb + (lane>>4)*4 + r][m = mb + (lane&15)]
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = m0 + wm * WM + j * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n0 + wn * WN + i * 16 + (lane >> 4) * 4;
      if (n >= N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (EPI & VLM_EPI_BIAS) {
        const uint2 b = *reinterpret_cast<const uint2*>(bias + n);
        v[0] += bf_lo(b.x); v[1] += bf_hi(b.x); v[2] += bf_lo(b.y); v[3] += bf_hi(b.y);
      }
      if (EPI & VLM_EPI_SWIGLU) {
        // interleaved (gate, up) rows of W -> N/2 outputs
        const float o0 = swiglu_(rbf(v[0]), rbf(v[1])), o1 = swiglu_(rbf(v[2]), rbf(v[3]));
        *reinterpret_cast<uint32_t*>(C + (size_t)m * ldc + (n >> 1)) = pack_bf2(o0, o1);
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
      if (EPI & VLM_EPI_RESIDUAL) {
        const uint2 rr = *reinterpret_cast<const uint2*>(res + (size_t)m * ldres + n);
        v[0] = rbf(v[0]) + bf_lo(rr.x); v[1] = rbf(v[1]) + bf_hi(rr.x);
        v[2] = rbf(v[2]) + bf_lo(rr.y); v[3] = rbf(v[3]) + bf_hi(rr.y);
      }
      uint2 o;
      o.x = pack_bf2(v[0], v[1]);
      o.y = pack_bf2(v[2], v[3]);
      *reinterpret_cast<uint2*>(C + (size_t)m * ldc + n) = o;
    }
  }
}

template <int BM, int BN, int EPI>
int launch_cfg(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
               int ldw, int ldc, int ldres, hipStream_t st) {
  const int tiles_m = vlm_cdiv(M, BM), tiles_n = vlm_cdiv(N, BN), nwg = tiles_m * tiles_n;
  const size_t lds = 2 * (size_t)(BM + BN) * ROWB;
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, EPI>), dim3(nwg), dim3(256), lds, st, (const bf16_t*)A, (const bf16_t*)W,
                     (const bf16_t*)bias, (const bf16_t*)res, (bf16_t*)C, M, N, K, lda, ldw, ldc, ldres, tiles_n, nwg);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

template <int EPI>
int launch_epi(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N, int K, int lda,
               int ldw, int ldc, int ldres, hipStream_t st) {
  // pick the largest tile that still yields >= ~1 workgroup per CU (256 CUs)
  const long t128 = (long)vlm_cdiv(M, 128) * vlm_cdiv(N, 128);
  const long t64n = (long)vlm_cdiv(M, 64) * vlm_cdiv(N, 128);
  if (t128 >= 200) return launch_cfg<128, 128, EPI>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
  if (t64n >= 200) return launch_cfg<64, 128, EPI>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
  return launch_cfg<64, 64, EPI>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st);
}

}  // namespace

extern "C" int vlm_gemm_bf16(const void* A, const void* W, const void* bias, const void* res, void* C, int M, int N,
                             int K, int lda, int ldw, int ldc, int ldres, int epilogue, void* stream) {
  if (!A || !W || !C || M < 0 || N <= 0 || K <= 0) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_BIAS) && !bias) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_RESIDUAL) && !res) return VLM_ERR_ARG;
  if (K % 8 != 0 || N % 8 != 0 || lda % 8 != 0 || ldw % 8 != 0 || ldc % 4 != 0) return VLM_ERR_SHAPE;
  if ((epilogue & VLM_EPI_RESIDUAL) && ldres % 4 != 0) return VLM_ERR_SHAPE;
  if (M == 0) return VLM_OK;
  hipStream_t st = (hipStream_t)stream;
#define GO(E) return launch_epi<E>(A, W, bias, res, C, M, N, K, lda, ldw, ldc, ldres, st)
  switch (epilogue) {
    case VLM_EPI_NONE: GO(VLM_EPI_NONE);
    case VLM_EPI_BIAS: GO(VLM_EPI_BIAS);
    case VLM_EPI_BIAS | VLM_EPI_GELU_FAST: GO(VLM_EPI_BIAS | VLM_EPI_GELU_FAST);
    case VLM_EPI_BIAS | VLM_EPI_GELU_ERF: GO(VLM_EPI_BIAS | VLM_EPI_GELU_ERF);
    case VLM_EPI_BIAS | VLM_EPI_RESIDUAL: GO(VLM_EPI_BIAS | VLM_EPI_RESIDUAL);
    case VLM_EPI_RESIDUAL: GO(VLM_EPI_RESIDUAL);
    case VLM_EPI_SWIGLU: GO(VLM_EPI_SWIGLU);
    default: return VLM_ERR_ARG;
  }
#undef GO
}
