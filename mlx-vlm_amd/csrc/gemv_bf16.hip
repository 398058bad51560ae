// Decode-time weight-streaming GEMV for gfx950 (HBM-bound):
//     y[m][n] = epilogue( sum_k x[m][k] * W[n][k] ),   m < MB <= 8
//
// Replaces the nn.Linear calls of the reference's per-token decode step
// (mlx_vlm/models/qwen2_vl/language.py:52-55,76,120 q/k/v/o projections;
//  mlp.py:9-14 + activations.py:7-9 SwiGLU MLP; language.py:514-517 lm_head /
//  embed_tokens.as_linear) with, optionally, the preceding nn.RMSNorm
//  (language.py:130-133,149-153,200) fused in as a prologue and the residual
//  add (language.py:151-153) / SiLU-gated product fused in as epilogues.
//
// Design (CDNA4): no LDS round trip for the weights - each lane streams 16-byte
// (8 x bf16) non-temporal loads of R weight rows straight into VGPRs while the
// activation chunk is read once per k-step (L1/L2 resident, or LDS when the
// RMSNorm prologue produced it), v_dot2c_f32_bf16 accumulates in fp32, and the
// R x MB partial sums are reduced with wavefront xor-shuffles.  One wave owns R
// consecutive output rows, so epilogues (bias, residual, swiglu on interleaved
// gate/up rows) are race-free and in-place safe.
#include "common.cuh"
#include "../../include/vlm_hip.h"

namespace {

__device__ __forceinline__ float dot8(const uint4& w, const uint4& x, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&w.x), *reinterpret_cast<const bf16x2_t*>(&x.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&w.y), *reinterpret_cast<const bf16x2_t*>(&x.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&w.z), *reinterpret_cast<const bf16x2_t*>(&x.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const bf16x2_t*>(&w.w), *reinterpret_cast<const bf16x2_t*>(&x.w), acc, false);
  return acc;
}

template <int R, int MB, bool NORM, int EPI>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                   const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                   const bf16_t* __restrict__ norm_w, bf16_t* __restrict__ y, int N, int K,
                                                   int ldx, int ldw, int ldy, int ldres, float eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunk = K >> 3;

  if (NORM) {
    // prologue: xs[m][:] = T(w * T(x[m] * rsqrt(mean(x^2) + eps)))  -> LDS (bf16)
    const uint4* wr = reinterpret_cast<const uint4*>(norm_w);
    for (int m = wave; m < MB; m += 4) {
      const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)m * ldx);
      float s = 0.f;
      for (int c = lane; c < nchunk; c += 64) {
        const uint4 u = xr[c];
        const float v[8] = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y), bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j] * v[j];
      }
      const float inv = rsqrtf(wave_sum(s) / (float)K + eps);
      uint4* xs = reinterpret_cast<uint4*>(smem + (size_t)m * K * 2);
      for (int c = lane; c < nchunk; c += 64) {
        const uint4 u = xr[c], wu = wr[c];
        uint4 o;
        o.x = pack_bf2(bf_lo(wu.x) * rbf(bf_lo(u.x) * inv), bf_hi(wu.x) * rbf(bf_hi(u.x) * inv));
        o.y = pack_bf2(bf_lo(wu.y) * rbf(bf_lo(u.y) * inv), bf_hi(wu.y) * rbf(bf_hi(u.y) * inv));
        o.z = pack_bf2(bf_lo(wu.z) * rbf(bf_lo(u.z) * inv), bf_hi(wu.z) * rbf(bf_hi(u.z) * inv));
        o.w = pack_bf2(bf_lo(wu.w) * rbf(bf_lo(u.w) * inv), bf_hi(wu.w) * rbf(bf_hi(u.w) * inv));
        xs[c] = o;
      }
    }
    __syncthreads();
  }

  const int row0 = (blockIdx.x * 4 + wave) * R;
  if (row0 >= N) return;
  const uint4* wrow[R];
#pragma unroll
  for (int r = 0; r < R; ++r) wrow[r] = reinterpret_cast<const uint4*>(W + (size_t)min(row0 + r, N - 1) * ldw);

  float acc[R][MB];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;

#pragma unroll 2
  for (int c = lane; c < nchunk; c += 64) {
    uint4 wv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wv[r] = nt_load16(wrow[r] + c);
    uint4 xv[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      if (NORM) xv[m] = *reinterpret_cast<const uint4*>(smem + ((size_t)m * K + (size_t)c * 8) * 2);
      else xv[m] = *reinterpret_cast<const uint4*>(x + (size_t)m * ldx + (size_t)c * 8);
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int m = 0; m < MB; ++m) acc[r][m] = dot8(wv[r], xv[m], acc[r][m]);
  }

#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[r][m] = wave_sum(acc[r][m]);

  if (EPI & VLM_EPI_SWIGLU) {
    // rows (2j, 2j+1) of W are (gate_j, up_j); R is even
#pragma unroll
    for (int r = 0; r < R; r += 2)
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        if (lane == (r >> 1) * MB + m && row0 + r + 1 < N) {
          const float o = swiglu_(rbf(acc[r][m]), rbf(acc[r + 1][m]));
          y[(size_t)m * ldy + ((row0 + r) >> 1)] = f2bf(o);
        }
      }
    return;
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      if (lane == r * MB + m && row0 + r < N) {
        float v = acc[r][m];
        if (EPI & VLM_EPI_BIAS) v += bf2f(bias[row0 + r]);
        if (EPI & VLM_EPI_RESIDUAL) v = rbf(v) + bf2f(res[(size_t)m * ldres + row0 + r]);
        y[(size_t)m * ldy + row0 + r] = f2bf(v);
      }
    }
}

template <int R, int MB, bool NORM, int EPI>
int launch(const void* x, const void* W, const void* bias, const void* res, const void* norm_w, void* y, int N, int K,
           int ldx, int ldw, int ldy, int ldres, float eps, hipStream_t st) {
  const int grid = vlm_cdiv(N, 4 * R);
  const size_t lds = NORM ? (size_t)MB * K * 2 : 0;
  hipLaunchKernelGGL((gemv_kernel<R, MB, NORM, EPI>), dim3(grid), dim3(256), lds, st, (const bf16_t*)x, (const bf16_t*)W,
                     (const bf16_t*)bias, (const bf16_t*)res, (const bf16_t*)norm_w, (bf16_t*)y, N, K, ldx, ldw, ldy, ldres, eps);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VLM_OK : VLM_ERR_HIP + (int)e;
}

template <int MB, bool NORM, int EPI>
int launch_r(const void* x, const void* W, const void* bias, const void* res, const void* norm_w, void* y, int N, int K,
             int ldx, int ldw, int ldy, int ldres, float eps, hipStream_t st) {
  // enough rows per wave to amortise the x reads, but keep >= ~2 workgroups per CU
  if (N >= 8192 && MB <= 4) return launch<4, MB, NORM, EPI>(x, W, bias, res, norm_w, y, N, K, ldx, ldw, ldy, ldres, eps, st);
  return launch<2, MB, NORM, EPI>(x, W, bias, res, norm_w, y, N, K, ldx, ldw, ldy, ldres, eps, st);
}

template <bool NORM, int EPI>
int launch_m(int M, const void* x, const void* W, const void* bias, const void* res, const void* norm_w, void* y, int N,
             int K, int ldx, int ldw, int ldy, int ldres, float eps, hipStream_t st) {
  switch (M) {
    case 1: return launch_r<1, NORM, EPI>(x, W, bias, res, norm_w, y, N, K, ldx, ldw, ldy, ldres, eps, st);
    case 2: return launch_r<2, NORM, EPI>(x, W, bias, res, norm_w, y, N, K, ldx, ldw, ldy, ldres, eps, st);
    case 3: case 4: {
      // M==3 runs as 4 with the caller's buffers padded to 4 rows
      if (M == 3) return VLM_ERR_SHAPE;
      return launch_r<4, NORM, EPI>(x, W, bias, res, norm_w, y, N, K, ldx, ldw, ldy, ldres, eps, st);
    }
    case 8: return launch_r<8, NORM, EPI>(x, W, bias, res, norm_w, y, N, K, ldx, ldw, ldy, ldres, eps, st);
    default: return VLM_ERR_SHAPE;
  }
}

}  // namespace

extern "C" int vlm_gemv_bf16(const void* x, const void* W, const void* bias, const void* res, const void* norm_w,
                             void* y, int M, int N, int K, int ldx, int ldw, int ldy, int ldres, float eps,
                             int epilogue, void* stream) {
  if (!x || !W || !y || N <= 0 || K <= 0) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_BIAS) && !bias) return VLM_ERR_ARG;
  if ((epilogue & VLM_EPI_RESIDUAL) && !res) return VLM_ERR_ARG;
  if (K % 8 != 0 || ldx % 8 != 0 || ldw % 8 != 0) return VLM_ERR_SHAPE;
  if ((epilogue & VLM_EPI_SWIGLU) && (N % 2 != 0)) return VLM_ERR_SHAPE;
  if (norm_w && (size_t)M * K * 2 > 64 * 1024) return VLM_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
#define GO(NORMV, E) return launch_m<NORMV, E>(M, x, W, bias, res, norm_w, y, N, K, ldx, ldw, ldy, ldres, eps, st)
  if (norm_w) {
    switch (epilogue) {
      case VLM_EPI_NONE: GO(true, VLM_EPI_NONE);
      case VLM_EPI_BIAS: GO(true, VLM_EPI_BIAS);
      case VLM_EPI_SWIGLU: GO(true, VLM_EPI_SWIGLU);
      default: return VLM_ERR_ARG;
    }
  } else {
    switch (epilogue) {
      case VLM_EPI_NONE: GO(false, VLM_EPI_NONE);
      case VLM_EPI_BIAS: GO(false, VLM_EPI_BIAS);
      case VLM_EPI_RESIDUAL: GO(false, VLM_EPI_RESIDUAL);
      case VLM_EPI_SWIGLU: GO(false, VLM_EPI_SWIGLU);
      default: return VLM_ERR_ARG;
    }
  }
#undef GO
}
