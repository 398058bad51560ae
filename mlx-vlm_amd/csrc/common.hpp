// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wave = 64 lanes everywhere; bf16 is handled as raw 16-bit words with
// round-to-nearest-even conversion (v_cvt_pk_bf16_f32 on gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // storage type of a bf16 element
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// 16-byte non-temporal (streaming) load: weights that ONE CU reads once (decode GEMV)
__device__ __forceinline__ uint4 nt_load16(const void* p) {
  const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

#define VLM_OK 0
#define VLM_ERR_ARG 1
#define VLM_ERR_SHAPE 2
#define VLM_ERR_HIP 1000

#define VLM_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return VLM_ERR_HIP + (int)e__;    \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// fp32 -> bf16, round to nearest even (NaN preserved) - same rule as torch/MLX.
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}
// two fp32 -> packed bf16x2 in ONE v_cvt_pk_bf16_f32 (the scalar casts cost a convert + shift + or each)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef float f32x2_t_ __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2v_t_ __attribute__((ext_vector_type(2)));
  const f32x2_t_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v_t_));
}
// round a float through bf16 (typed-graph emulation of a materialised T tensor)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// Wave-wide reductions on the DPP data path.  __shfl_xor compiles to ds_bpermute_b32 on gfx950 - an LDS-crossbar round
// trip of ~100 cycles per step, six dependent steps per reduction, on the critical tail of every GEMV launch (found in the
// .s of gemv_rowwave_kernel).  DPP operands ride inside the VALU instruction (v_add_f32_dpp): quad swaps, row_half_mirror
// and row_mirror leave every 16-lane row with its row sum; row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3) carry
// the row sums up so that lane 63 holds the total, which v_readlane_b32 broadcasts through an SGPR.  Result in EVERY lane
// (as the xor butterfly gave); fp32 summation order differs from the butterfly's.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float vlm_dpp(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += vlm_dpp<0xB1, 0xf>(0.f, v);     // quad_perm [1,0,3,2]
  v += vlm_dpp<0x4E, 0xf>(0.f, v);     // quad_perm [2,3,0,1]
  v += vlm_dpp<0x141, 0xf>(0.f, v);    // row_half_mirror
  v += vlm_dpp<0x140, 0xf>(0.f, v);    // row_mirror
  v += vlm_dpp<0x142, 0xa>(0.f, v);    // row_bcast:15 into rows 1 and 3
  v += vlm_dpp<0x143, 0xc>(0.f, v);    // row_bcast:31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, vlm_dpp<0xB1, 0xf>(v, v));
  v = fmaxf(v, vlm_dpp<0x4E, 0xf>(v, v));
  v = fmaxf(v, vlm_dpp<0x141, 0xf>(v, v));
  v = fmaxf(v, vlm_dpp<0x140, 0xf>(v, v));
  v = fmaxf(v, vlm_dpp<0x142, 0xa>(v, v));
  v = fmaxf(v, vlm_dpp<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// lane l <-> lane l ^ 16 / l ^ 32 exchanges as ONE gfx950 instruction each (v_permlane16_swap_b32 / v_permlane32_swap_b32:
// with both operands = v, the results are [r0 r0 r2 r2] | [r1 r1 r3 r3] resp. [lo lo] | [hi hi]) instead of a ds_bpermute.
// COMPILER TRAP (hipcc / ROCm 7.2; the one gemv_bf16.hip documents): __builtin_bit_cast applied to a vector ELEMENT
// (r[1]) silently reads element 0 - found in the .s as `v_add_f32 v2, v66, v66` after the swap, and on the GPU as wrong
// attention.  The elements are copied to scalars first.
__device__ __forceinline__ void vlm_xor16_pair(float v, float& a, float& b) {
  typedef unsigned int u32x2_t_ __attribute__((ext_vector_type(2)));
  const u32x2_t_ r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void vlm_xor32_pair(float v, float& a, float& b) {
  typedef unsigned int u32x2_t_ __attribute__((ext_vector_type(2)));
  const u32x2_t_ r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
// reductions over the 4 lanes {l & 15, + 16, + 32, + 48} (one MFMA column held by the four 16-lane rows)
__device__ __forceinline__ float col4_max(float v) {
  float a, b;
  vlm_xor16_pair(v, a, b);
  v = fmaxf(a, b);
  vlm_xor32_pair(v, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float col4_sum(float v) {
  float a, b;
  vlm_xor16_pair(v, a, b);
  v = a + b;
  vlm_xor32_pair(v, a, b);
  return a + b;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` = >=16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// sigmoid on the hardware transcendental units: v_exp_f32 + v_rcp_f32 (~1 ulp each; the result is rounded to bf16
// by the caller).  NOT __frcp_rn / 1.0f/x: those compile to the IEEE division sequence (v_div_scale, v_div_fmas,
// v_div_fixup + 4 fma, found in the GEMM epilogue's .s) and made the bias+GELU epilogue cost a quarter of the tile.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
// The activations follow the reference's typed graph (pinned by running the reference's files over
// oracle/mlx_shim, tests/test_oracle_ref_golden.py): every elementary MLX op rounds to bf16 and a python scalar is
// converted to bf16 BEFORE the op (weak typing), so 1.702 is 1.703125 and sqrt(2) is 1.4140625 here.
// x is the bf16-rounded linear output; the caller rounds the returned value once more.
// nn.GELU(approx="fast") : x * sigmoid(1.702 x)        (reference vision.py:167)
__device__ __forceinline__ float gelu_fast_(float x) {
  const float t = rbf(1.703125f * x);
  return x * rbf(sigmoidf_(t));
}
// The same typed graph on TWO elements at once (the GEMM epilogues, where this is 1770 VALU instructions per lane and
// 256-wide tile: profiles/r05_gemm256_workgroup_timeline.txt - 8.8 of a workgroup's 40.6 us): every rounding to bf16 is ONE
// v_cvt_pk_bf16_f32 for the pair (the scalar form spent a convert on each element), the multiplies and the add are
// v_pk_mul_f32 / v_pk_add_f32; the two v_exp_f32 + two v_rcp_f32 stay.  Bit-identical to gelu_fast_ per element (same
// operations, same rounding points, IEEE multiply / add either way).  x0, x1: the bf16-rounded linear outputs.
typedef float vlm_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vlm_f32x2_t rbf2(vlm_f32x2_t v) {
  const uint32_t p = pack_bf2(v[0], v[1]);
  return vlm_f32x2_t{bf_lo(p), bf_hi(p)};
}
__device__ __forceinline__ vlm_f32x2_t gelu_fast2_(vlm_f32x2_t x) {
  const vlm_f32x2_t t = rbf2(x * 1.703125f);
  const vlm_f32x2_t y = t * -1.44269504088896340736f;
  vlm_f32x2_t e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
  e = e + 1.0f;
  const vlm_f32x2_t sg = rbf2(vlm_f32x2_t{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])});
  return x * sg;
}
// nn.GELU() : x * (1 + erf(x / sqrt(2))) / 2            (reference vision.py:112)
__device__ __forceinline__ float gelu_erf_(float x) {
  const float t = rbf(x / 1.4140625f);
  const float o = rbf(1.0f + rbf(erff(t)));
  return 0.5f * rbf(x * o);
}
// swiglu typed graph: sig -> T, g*sig -> T, *u -> T      (reference activations.py:7-9)
__device__ __forceinline__ float swiglu_(float g, float u) {
  float sig = rbf(sigmoidf_(g));
  float silu = rbf(g * sig);
  return silu * u;
}

// V pool key-slot order inside a 64-token page (see attn_decode.hip): keys of each 32-block are stored in
// the k-slot order of the P.V MFMA so that a V^T operand fragment is one contiguous 16-byte load.
__host__ __device__ __forceinline__ int vlm_vslot(int within) {
  const int kk = within & 31;
  return (within & 32) + 8 * ((kk & 15) >> 2) + 4 * (kk >> 4) + (kk & 3);
}

// Internal GEMM epilogue id: 2-D rotary embedding of the vision attention fused into the qkv projection
// (apply_rotary_pos_emb_vision, reference mlx_vlm/models/qwen2_vl/vision.py:35-50,141-142).  The rotation pairs
// element d with d + hd/2 of a head; the checkpoint's q / k rows are INTERLEAVED at load ((d, d + hd/2) -> columns
// (2j, 2j + 1); q.k is invariant under a common permutation of the head dimension), so a pair sits in one 16-byte store
// chunk of the epilogue.  Conventions inside the GEMM launch chain: `res` = fp32 table [2][M][hd/2] (cos rows, then sin
// rows), `ldres` = hd | rope_cols << 12 (columns n < rope_cols are q / k heads).
#define VLM_EPI_ROPE2D 32
__device__ __forceinline__ uint4 vlm_rope2d_chunk(uint4 u, const void* table, int M, int m, int n, int packed) {
  const int hd = packed & 0xfff, rope_cols = packed >> 12;
  if (n >= rope_cols) return u;
  const int half = hd >> 1, p0 = (n % hd) >> 1;
  const float* t = static_cast<const float*>(table);
  const float4 c = *reinterpret_cast<const float4*>(t + (size_t)m * half + p0);
  const float4 s = *reinterpret_cast<const float4*>(t + ((size_t)M + m) * half + p0);
  // x is the bf16-rounded linear output (bias included); fp32 rotation, one rounding (vision.py:46-50)
  const float a0 = bf_lo(u.x), b0 = bf_hi(u.x), a1 = bf_lo(u.y), b1 = bf_hi(u.y);
  const float a2 = bf_lo(u.z), b2 = bf_hi(u.z), a3 = bf_lo(u.w), b3 = bf_hi(u.w);
  u.x = pack_bf2(a0 * c.x - b0 * s.x, b0 * c.x + a0 * s.x);
  u.y = pack_bf2(a1 * c.y - b1 * s.y, b1 * c.y + a1 * s.y);
  u.z = pack_bf2(a2 * c.z - b2 * s.z, b2 * c.z + a2 * s.z);
  u.w = pack_bf2(a3 * c.w - b3 * s.w, b3 * c.w + a3 * s.w);
  return u;
}

static inline int vlm_cdiv(int a, int b) { return (a + b - 1) / b; }
