// What does ds_read_b64_tr_b16 return?  LDS holds lds[i] = i; lane l passes the address of elements [A(l), A(l)+4).
// Two address patterns: (a) A = 4 l (lane-linear), (b) a [4 rows][16 cols] block per 16-lane group at a row pitch of 88
// elements: lane i -> row (i & 15) / 4, cols 4 (i & 3).  Prints the 4 values each lane receives.
#include <hip/hip_runtime.h>

#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  const int a = mode == 0 ? 4 * l : (g * 4 + i / 4) * 88 + 4 * (i & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short *d, h[256];
  hipMalloc(&d, 512);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "");
  }
  return 0;
}
