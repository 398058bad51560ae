// Standalone driver for the second skinny-M decode GEMM (csrc/gemv_mfma2.hip) on MI355X: one projection shape, timing over
// rotating weight copies (HIP events around a captured graph of launches) and, in a -DMFMA2_STAMPS=<block> build, the
// wall-clock phase stamps of that workgroup's wave 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form [-DVLM_MFMA2_W4_TU] [-DMFMA2_STAMPS=0] \
//         scripts/mfma2_probe.hip -Iinclude -o scripts/bin/mfma2_probe[_w4]
//   scripts/bin/mfma2_probe N K M epilogue(0 none, 8 residual, 16 swiglu) norm(0/1) [reps=40]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../mlx-vlm_amd/csrc/norm.hip"
#include "../mlx-vlm_amd/csrc/gemv_mfma2.hip"

#define CK(x)                                                                                 \
  do {                                                                                        \
    hipError_t e__ = (x);                                                                     \
    if (e__ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
      exit(2);                                                                                \
    }                                                                                         \
  } while (0)

__global__ void fill_u32(uint32_t* p, size_t n, uint32_t seed, uint32_t and_mask, uint32_t or_mask) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 0x9E3779B1u + seed;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    p[i] = (x & and_mask) | or_mask;
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 16384, K = argc > 2 ? atoi(argv[2]) : 3072, M = argc > 3 ? atoi(argv[3]) : 16;
  const int epi = argc > 4 ? atoi(argv[4]) : 16, norm = argc > 5 ? atoi(argv[5]) : 1, reps = argc > 6 ? atoi(argv[6]) : 40;
#ifdef VLM_MFMA2_W4_TU
  const bool w4 = true;
#else
  const bool w4 = false;
#endif
  const size_t wbytes = w4 ? (size_t)N * K / 2 : (size_t)N * K * 2, sbbytes = w4 ? (size_t)N * (K / 64) * 4 : 0;
  const int ncopy = (int)std::max<size_t>(2, std::min<size_t>(24, (size_t)1.2e9 / (wbytes + sbbytes) + 1));
  std::vector<void*> W(ncopy), SB(ncopy, nullptr);
  for (int c = 0; c < ncopy; ++c) {
    CK(hipMalloc(&W[c], wbytes));
    // bf16: small random values (sign | exponent 2^-6.. | mantissa); 4-bit: random nibbles
    hipLaunchKernelGGL(fill_u32, dim3(4096), dim3(256), 0, 0, (uint32_t*)W[c], wbytes / 4, 77u + c, w4 ? 0xffffffffu : 0x807f807fu,
                       w4 ? 0u : 0x3c003c00u);
    if (w4) {
      CK(hipMalloc(&SB[c], sbbytes));
      hipLaunchKernelGGL(fill_u32, dim3(1024), dim3(256), 0, 0, (uint32_t*)SB[c], sbbytes / 4, 99u + c, 0x000f000fu, 0xbd003b80u);
    }
  }
  void *x, *y, *res, *nw, *ws;
  CK(hipMalloc(&x, (size_t)16 * K * 2));
  CK(hipMalloc(&nw, (size_t)K * 2));
  CK(hipMalloc(&y, (size_t)16 * N * 2));
  CK(hipMalloc(&res, (size_t)16 * N * 2));
  const size_t wsb = VLM_MFMA_WS_XN_OFFSET + VLM_MFMA_WS_XN_BYTES;
  CK(hipMalloc(&ws, wsb));
  CK(hipMemset(ws, 0, wsb));
  hipLaunchKernelGGL(fill_u32, dim3(256), dim3(256), 0, 0, (uint32_t*)x, (size_t)16 * K / 2, 5u, 0x807f807fu, 0x3f003f00u);
  hipLaunchKernelGGL(fill_u32, dim3(256), dim3(256), 0, 0, (uint32_t*)nw, (size_t)K / 2, 6u, 0x000f000fu, 0x3f803f80u);
  hipLaunchKernelGGL(fill_u32, dim3(256), dim3(256), 0, 0, (uint32_t*)res, (size_t)16 * N / 2, 7u, 0x807f807fu, 0x3f003f00u);
  CK(hipDeviceSynchronize());
  hipStream_t st;
  CK(hipStreamCreate(&st));
  auto run = [&](int c) {
    const int rc = VLM_MFMA2_ENTRY(x, W[c], SB[c], nullptr, (epi & 8) ? res : nullptr, norm ? nw : nullptr, y, M, N, K, K, w4 ? 8 : K,
                                   (epi & 16) ? N / 2 : N, N, 1e-6f, epi, nullptr, ws, st);
    if (rc) { fprintf(stderr, "rc=%d\n", rc); exit(3); }
  };
  for (int c = 0; c < ncopy; ++c) run(c);
  CK(hipStreamSynchronize(st));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int r = 0; r < reps; ++r) run(r % ncopy);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int t = 0; t < 3; ++t) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms * 1e3f / reps);
  }
  std::vector<uint16_t> hy(256);
  CK(hipMemcpy(hy.data(), y, 512, hipMemcpyDeviceToHost));
  uint64_t cs = 0;
  for (auto v : hy) cs = cs * 1315423911u + v;
  printf("%s N=%d K=%d M=%d epi=%d norm=%d: %.2f us per projection (%s), %.2f TB/s, checksum %016llx\n", w4 ? "w4" : "bf16", N, K, M, epi,
         norm, best, norm ? "incl. the rows kernel when K is split" : "one launch", (wbytes + sbbytes) / best * 1e-6, (unsigned long long)cs);
#ifdef MFMA2_STAMPS
  run(0);
  CK(hipStreamSynchronize(st));
  unsigned long long h[64];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mfma2_stamps), sizeof(h)));
  printf("workgroup %d wave 0, us since its start: prologue done %.2f |", MFMA2_STAMPS, (h[1] - h[0]) * 0.01);
  for (int k = 0; k < 6 && h[2 + 3 * k]; ++k)
    printf(" unit %d: chunks %.2f barrier %.2f epilogue %.2f |", k, (h[2 + 3 * k] - h[0]) * 0.01, (h[3 + 3 * k] - h[0]) * 0.01,
           (h[4 + 3 * k] - h[0]) * 0.01);
  printf(" end %.2f\n", (h[62] - h[0]) * 0.01);
#endif
  return 0;
}
