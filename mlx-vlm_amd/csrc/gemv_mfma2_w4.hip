// The 4-bit (MLX affine, group 64) instantiations of the second skinny-M decode GEMM as a translation unit of their own
// (compiled in parallel with the bf16 ones).
#define VLM_MFMA2_W4_TU 1
#include "gemv_mfma2.hip"
