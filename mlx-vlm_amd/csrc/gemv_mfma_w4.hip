// The MLX affine 4-bit instantiations of the skinny-M decode GEMM (csrc/gemv_mfma.hip, W4 = true: vlm_gemv_mfma_try_w4) as their own
// translation unit, so that the ~120 kernels of each weight format compile in parallel.
#define VLM_MFMA_W4_TU 1
#include "gemv_mfma.hip"
