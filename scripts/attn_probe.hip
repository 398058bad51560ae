// Standalone driver for the ViT attention kernel on MI355X (no torch: starts in milliseconds).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DATTN_STAMPS=<blockIdx.x>] scripts/attn_probe.hip -Iinclude -o scripts/bin/attn_probe
//   scripts/bin/attn_probe [images=16] [seq=576] [heads=16] [reps=50]
// bench.py's ViT workload: 16 x 336x336 images -> 16 segments of 576 patches, 16 heads of 80.  Prints the average launch
// time, the achieved TFLOP/s (4 * L^2 * 80 flops per head and segment), an output checksum (A/B of variants must agree bit
// for bit) and, in a -DATTN_STAMPS build, the per-phase cycle sums of one workgroup's four waves.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../mlx-vlm_amd/csrc/attn_prefill.hip"

#define CK(x)                                                                                 \
  do {                                                                                        \
    hipError_t e__ = (x);                                                                     \
    if (e__ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
      exit(2);                                                                                \
    }                                                                                         \
  } while (0)

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 0x9E3779B1u + seed;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    const float v = 2.0f * scale * (((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f) - 0.5f);
    uint32_t b;
    memcpy(&b, &v, 4);
    b += 0x7fffu + ((b >> 16) & 1u);
    p[i] = (uint16_t)(b >> 16);
  }
}

int main(int argc, char** argv) {
  const int images = argc > 1 ? atoi(argv[1]) : 16, L = argc > 2 ? atoi(argv[2]) : 576, H = argc > 3 ? atoi(argv[3]) : 16;
  const int reps = argc > 4 ? atoi(argv[4]) : 50, D = 80, E = H * D, T = images * L;
  uint16_t *qkv, *out;
  int* cu;
  CK(hipMalloc(&qkv, (size_t)T * 3 * E * 2));
  CK(hipMalloc(&out, (size_t)T * E * 2));
  CK(hipMalloc(&cu, (images + 1) * 4));
  std::vector<int> h(images + 1);
  for (int i = 0; i <= images; ++i) h[i] = i * L;
  CK(hipMemcpy(cu, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, qkv, (size_t)T * 3 * E, 12345u, 1.5f);
  CK(hipDeviceSynchronize());
  const int nqb = images * ((L + 127) / 128);
  auto run = [&]() {
    int rc = vlm_attn_prefill(qkv, qkv + E, qkv + 2 * E, out, 3 * E, 3 * E, 3 * E, E, cu, images, nqb, H, H, D, 1.0f / sqrtf((float)D),
                              2, nullptr);
    if (rc) { fprintf(stderr, "rc=%d\n", rc); exit(3); }
  };
  for (int i = 0; i < 5; ++i) run();
  CK(hipDeviceSynchronize());
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e30f, tot = 0.f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
    tot += ms;
  }
  const double us = best * 1e3 / reps, flops = 4.0 * L * L * D * H * images;
  std::vector<uint16_t> ho((size_t)T * E);
  CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
  uint64_t sum = 0;
  for (size_t i = 0; i < ho.size(); ++i) sum = sum * 1000003ull + ho[i];
  printf("attn_prefill<80> %d x %d x %d heads: %.2f us/launch (best of 5 x %d; mean %.2f), %.1f TFLOP/s, checksum %016llx\n", images,
         L, H, us, reps, tot * 1e3 / (5 * reps), flops / us * 1e-6, (unsigned long long)sum);
#ifdef ATTN_STAMPS
  unsigned long long st[4][8];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_attn_stamps), sizeof(st)));
  const char* names[6] = {"issue next tile's global loads", "QK^T MFMAs issued", "softmax + P pack", "P.V MFMAs issued",
                          "staging regs -> LDS (vmcnt wait)", "barrier"};
  const int ntiles = (L + 63) / 64;
  printf("phase cycle sums over %d tiles, workgroup blockIdx.x=%d (s_memtime ticks; per tile in brackets)\n", ntiles, ATTN_STAMPS);
  for (int i = 0; i < 6; ++i) {
    printf("  %-36s", names[i]);
    for (int w = 0; w < 4; ++w) printf("  w%d %7llu [%5llu]", w, st[w][i], st[w][i] / ntiles);
    printf("\n");
  }
#endif
  return 0;
}
