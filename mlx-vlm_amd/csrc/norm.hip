// LayerNorm / RMSNorm for gfx950.  HBM-bound row kernels: one wave per row,
// 16-byte (8 x bf16) loads, the row kept in registers between the statistics
// pass and the normalise pass, fp32 statistics, wavefront-shuffle reductions.
//
// Replaces: nn.LayerNorm(eps=1e-6) -> mx.fast.layer_norm
//             (reference mlx_vlm/models/qwen2_vl/vision.py:109,180-181)
//           nn.RMSNorm -> mx.fast.rms_norm
//             (reference mlx_vlm/models/qwen2_vl/language.py:130-133,168)
//           and the residual adds of language.py:151-153 (fused variant).
#include "common.hpp"
#include "../../include/vlm_hip.h"

namespace {

// NCH = number of 512-element slabs per row a lane may hold (dim <= 512*NCH).
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                        int rows, int dim, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int nchunk = dim >> 3;
  const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * dim);
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      uint4 u = xr[c];
      v[i][0] = bf_lo(u.x); v[i][1] = bf_hi(u.x); v[i][2] = bf_lo(u.y); v[i][3] = bf_hi(u.y);
      v[i][4] = bf_lo(u.z); v[i][5] = bf_hi(u.z); v[i][6] = bf_lo(u.w); v[i][7] = bf_hi(u.w);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = v[i][j] - mean; q += d * d; }
    }
  }
  const float inv = rsqrtf(wave_sum(q) / (float)dim + eps);
  const uint4* wr = reinterpret_cast<const uint4*>(w);
  const uint4* br = reinterpret_cast<const uint4*>(b);
  uint4* yr = reinterpret_cast<uint4*>(y + (size_t)row * dim);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      uint4 wu = wr[c], bu = br[c], o;
      float wv[8] = {bf_lo(wu.x), bf_hi(wu.x), bf_lo(wu.y), bf_hi(wu.y), bf_lo(wu.z), bf_hi(wu.z), bf_lo(wu.w), bf_hi(wu.w)};
      float bv[8] = {bf_lo(bu.x), bf_hi(bu.x), bf_lo(bu.y), bf_hi(bu.y), bf_lo(bu.z), bf_hi(bu.z), bf_lo(bu.w), bf_hi(bu.w)};
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = (v[i][j] - mean) * inv * wv[j] + bv[j];
      o.x = pack_bf2(r[0], r[1]); o.y = pack_bf2(r[2], r[3]); o.z = pack_bf2(r[4], r[5]); o.w = pack_bf2(r[6], r[7]);
      yr[c] = o;
    }
  }
}

// y = w * T(h * rsqrt(mean(h^2)+eps)),  h = x (+ res);  optionally h_out = T(x + res).
template <int NCH, bool HAS_RES>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                      const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                      bf16_t* __restrict__ h_out, int rows, int dim, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int nchunk = dim >> 3;
  const uint4* xr = reinterpret_cast<const uint4*>(x + (size_t)row * dim);
  const uint4* rr = HAS_RES ? reinterpret_cast<const uint4*>(res + (size_t)row * dim) : nullptr;
  uint4* hr = (HAS_RES && h_out) ? reinterpret_cast<uint4*>(h_out + (size_t)row * dim) : nullptr;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      uint4 u = xr[c];
      v[i][0] = bf_lo(u.x); v[i][1] = bf_hi(u.x); v[i][2] = bf_lo(u.y); v[i][3] = bf_hi(u.y);
      v[i][4] = bf_lo(u.z); v[i][5] = bf_hi(u.z); v[i][6] = bf_lo(u.w); v[i][7] = bf_hi(u.w);
      if (HAS_RES) {
        uint4 r = rr[c];
        float rv[8] = {bf_lo(r.x), bf_hi(r.x), bf_lo(r.y), bf_hi(r.y), bf_lo(r.z), bf_hi(r.z), bf_lo(r.w), bf_hi(r.w)};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = rbf(v[i][j] + rv[j]);
        if (hr) {
          uint4 o;
          o.x = pack_bf2(v[i][0], v[i][1]); o.y = pack_bf2(v[i][2], v[i][3]);
          o.z = pack_bf2(v[i][4], v[i][5]); o.w = pack_bf2(v[i][6], v[i][7]);
          hr[c] = o;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j] * v[i][j];
    }
  }
  const float inv = rsqrtf(wave_sum(s) / (float)dim + eps);
  const uint4* wr = reinterpret_cast<const uint4*>(w);
  uint4* yr = reinterpret_cast<uint4*>(y + (size_t)row * dim);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nchunk) {
      uint4 wu = wr[c], o;
      float wv[8] = {bf_lo(wu.x), bf_hi(wu.x), bf_lo(wu.y), bf_hi(wu.y), bf_lo(wu.z), bf_hi(wu.z), bf_lo(wu.w), bf_hi(wu.w)};
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = wv[j] * rbf(v[i][j] * inv);
      o.x = pack_bf2(r[0], r[1]); o.y = pack_bf2(r[2], r[3]); o.z = pack_bf2(r[4], r[5]); o.w = pack_bf2(r[6], r[7]);
      yr[c] = o;
    }
  }
}

}  // namespace

template <int NCH>
void launch_ln(dim3 grid, hipStream_t st, const void* x, const void* w, const void* b, void* y, int rows, int dim, float eps) {
  hipLaunchKernelGGL((layernorm_kernel<NCH>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b,
                     (bf16_t*)y, rows, dim, eps);
}
template <int NCH>
void launch_rms(dim3 grid, hipStream_t st, const void* x, const void* res, const void* w, void* y, void* h_out, int rows,
                int dim, float eps) {
  if (res)
    hipLaunchKernelGGL((rmsnorm_kernel<NCH, true>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)res,
                       (const bf16_t*)w, (bf16_t*)y, (bf16_t*)h_out, rows, dim, eps);
  else
    hipLaunchKernelGGL((rmsnorm_kernel<NCH, false>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)nullptr,
                       (const bf16_t*)w, (bf16_t*)y, (bf16_t*)nullptr, rows, dim, eps);
}

extern "C" int vlm_layernorm(const void* x, const void* w, const void* b, void* y, int rows, int dim, float eps,
                             void* stream) {
  if (!x || !w || !b || !y || rows < 0 || dim <= 0) return VLM_ERR_ARG;
  if (dim % 8 != 0 || dim > 8192) return VLM_ERR_SHAPE;
  if (rows == 0) return VLM_OK;
  const int nch = vlm_cdiv(dim, 512);
  dim3 grid(vlm_cdiv(rows, 4));
  hipStream_t st = (hipStream_t)stream;
  if (nch <= 1) launch_ln<1>(grid, st, x, w, b, y, rows, dim, eps);
  else if (nch <= 2) launch_ln<2>(grid, st, x, w, b, y, rows, dim, eps);
  else if (nch <= 3) launch_ln<3>(grid, st, x, w, b, y, rows, dim, eps);
  else if (nch <= 4) launch_ln<4>(grid, st, x, w, b, y, rows, dim, eps);
  else if (nch <= 8) launch_ln<8>(grid, st, x, w, b, y, rows, dim, eps);
  else launch_ln<16>(grid, st, x, w, b, y, rows, dim, eps);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_rmsnorm_residual(const void* x, const void* res, const void* w, void* y, void* h_out, int rows,
                                    int dim, float eps, void* stream) {
  if (!x || !w || !y || rows < 0 || dim <= 0) return VLM_ERR_ARG;
  if (dim % 8 != 0 || dim > 8192) return VLM_ERR_SHAPE;
  if (rows == 0) return VLM_OK;
  const int nch = vlm_cdiv(dim, 512);
  dim3 grid(vlm_cdiv(rows, 4));
  hipStream_t st = (hipStream_t)stream;
  if (nch <= 1) launch_rms<1>(grid, st, x, res, w, y, h_out, rows, dim, eps);
  else if (nch <= 2) launch_rms<2>(grid, st, x, res, w, y, h_out, rows, dim, eps);
  else if (nch <= 3) launch_rms<3>(grid, st, x, res, w, y, h_out, rows, dim, eps);
  else if (nch <= 4) launch_rms<4>(grid, st, x, res, w, y, h_out, rows, dim, eps);
  else if (nch <= 8) launch_rms<8>(grid, st, x, res, w, y, h_out, rows, dim, eps);
  else launch_rms<16>(grid, st, x, res, w, y, h_out, rows, dim, eps);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
