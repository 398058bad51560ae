// Feasibility probe for a persistent decode kernel on gfx950 (not product code):
//   1. cost of a grid-wide barrier across 256 workgroups / 8 XCDs (flat counter vs per-group counters)
//   2. throughput of the "phased GEMV with register prefetch across barriers" pattern on Qwen2-VL-2B layer shapes
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mega_skel.hip -o scripts/mega_skel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float dot8(u32x4_t w, u32x4_t x, float a) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { a = fmaf(bf_lo(w[i]), bf_lo(x[i]), a); a = fmaf(bf_hi(w[i]), bf_hi(x[i]), a); }
  return a;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- grid barrier: monotonic counter, target = generation * nblocks
struct Bar { unsigned* flat; unsigned* grp; unsigned* top; int nblocks; int mode; };

__device__ __forceinline__ void grid_barrier(const Bar& b, unsigned& gen) {
  asm volatile("" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    ++gen;
    __threadfence();
    if (b.mode == 0) {
      __hip_atomic_fetch_add(b.flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = gen * (unsigned)b.nblocks;
      while (__hip_atomic_load(b.flat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    } else {
      // two-level: 8 groups (blockIdx & 7 ~ XCD under round-robin dispatch); the last arriver of a group bumps top
      const int g = blockIdx.x & 7, gsize = (b.nblocks + 7 - g) / 8;
      const unsigned old = __hip_atomic_fetch_add(b.grp + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == gen * (unsigned)gsize) __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = gen * 8u;
      while (__hip_atomic_load(b.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
  }
  __syncthreads();
  asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(512) void barrier_loop(Bar b, int iters, unsigned long long* cycles) {
  unsigned gen = 0;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; ++i) grid_barrier(b, gen);
  if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = wall_clock64() - t0;
}

// ---- phased streaming skeleton.  Per layer, per wave (2048 waves): qkv 3 chunks, o 3, gate/up 27, down 18
// (chunk = one 16-byte load per lane = 1 KiB per wave).  x is read with coherent (sc0 sc1) loads after each barrier.
constexpr int CQ = 3, CO = 3, CG = 24, CD = 15, CL = CQ + CO + CG + CD;   // 51 KiB per wave per layer

__device__ __forceinline__ u32x4_t ldw(const u32x4_t* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ u32x4_t ldx(const u32x4_t* p) {
  u32x4_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int PREFETCH>
__global__ __launch_bounds__(512) void phased(Bar b, const u32x4_t* __restrict__ W, int layers, unsigned short* act,
                                              float* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t gw = (size_t)blockIdx.x * 8 + wave, nw = (size_t)gridDim.x * 8;
  unsigned gen = 0;
  u32x4_t wq[CQ], wo[CO], wg[CG], wd[CD];
  const u32x4_t* xb = reinterpret_cast<const u32x4_t*>(act);
  auto base = [&](int l, int off) { return W + (((size_t)l * CL + off) * nw + gw) * 64 + (unsigned)lane; };   // uniform base + lane offset
  auto loadq = [&](int l) {
#pragma unroll
    for (int j = 0; j < CQ; ++j) wq[j] = ldw(base(l, j) + 0);
  };
  auto loado = [&](int l) {
#pragma unroll
    for (int j = 0; j < CO; ++j) wo[j] = ldw(base(l, CQ + j));
  };
  auto loadg = [&](int l) {
#pragma unroll
    for (int j = 0; j < CG; ++j) wg[j] = ldw(base(l, CQ + CO + j));
  };
  auto loadd = [&](int l) {
#pragma unroll
    for (int j = 0; j < CD; ++j) wd[j] = ldw(base(l, CQ + CO + CG + j));
  };
  float total = 0.f;
  if (PREFETCH) { loadq(0); loado(0); loadg(0); loadd(0); }
  for (int l = 0; l < layers; ++l) {
    const int ln = l + 1 < layers ? l + 1 : l;
    // P1 qkv
    {
      if (!PREFETCH) loadq(l);
      u32x4_t x[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) x[j] = ldx(xb + j * 64 + lane);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2])::"memory");
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < CQ; ++j) a = dot8(wq[j], x[j], a);
      a = wave_sum(a);
      if (PREFETCH) loadq(ln);
      if (lane == 0) act[4096 + gw] = (unsigned short)(__float_as_uint(a) >> 16);
      total += a;
      grid_barrier(b, gen);
    }
    // P2 attention stand-in: two workgroups do a dependent load chain, everyone else just passes
    {
      if (blockIdx.x < 2) {
        u32x4_t x0 = ldx(xb + 1024 + lane);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x0)::"memory");
        const u32x4_t* kv = W + (((size_t)l * CL) * nw + (x0[0] & 7)) * 64 + lane;
        u32x4_t k0 = ldx(kv);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(k0)::"memory");
        if (lane == 0) act[8192 + gw] = (unsigned short)k0[0];
      }
      grid_barrier(b, gen);
    }
    // P3 o_proj
    {
      if (!PREFETCH) loado(l);
      u32x4_t x[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) x[j] = ldx(xb + 512 + j * 64 + lane);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2])::"memory");
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < CO; ++j) a = dot8(wo[j], x[j], a);
      a = wave_sum(a);
      if (PREFETCH) loado(ln);
      if (lane == 0) act[12288 + gw] = (unsigned short)(__float_as_uint(a) >> 16);
      total += a;
      grid_barrier(b, gen);
    }
    // P4 gate/up: 9 rows x 3 chunks
    {
      if (!PREFETCH) loadg(l);
      u32x4_t x[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) x[j] = ldx(xb + 768 + j * 64 + lane);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2])::"memory");
      float a[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        a[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) a[r] = dot8(wg[r * 3 + j], x[j], a[r]);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) a[r] = wave_sum(a[r]);
      if (PREFETCH) loadg(ln);
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += a[r];
      if (lane < 5) act[16384 + gw * 5 + lane] = (unsigned short)(__float_as_uint(s) >> 16);
      total += s;
      grid_barrier(b, gen);
    }
    // P5 down: 1 row x 18 chunks, x in 6 pieces of 3
    {
      if (!PREFETCH) loadd(l);
      float a = 0.f;
#pragma unroll
      for (int p = 0; p < 5; ++p) {
        u32x4_t x[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) x[j] = ldx(xb + 2048 + (p * 3 + j) * 64 + lane);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2])::"memory");
#pragma unroll
        for (int j = 0; j < 3; ++j) a = dot8(wd[p * 3 + j], x[j], a);
      }
      a = wave_sum(a);
      if (PREFETCH) loadd(ln);
      if (lane == 0) act[gw & 1023] = (unsigned short)(__float_as_uint(a) >> 16);
      total += a;
      grid_barrier(b, gen);
    }
  }
  if (total == 123.456f) sink[0] = total;
}

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int ncu = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, ncu);
  unsigned* ctr;
  CK(hipMalloc(&ctr, 4096 * 4));
  unsigned long long* cyc;
  CK(hipMalloc(&cyc, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (int mode = 0; mode < 2; ++mode)
    for (int nb : {64, 128, 256}) {
      if (nb > ncu) continue;
      CK(hipMemsetAsync(ctr, 0, 4096 * 4, st));
      Bar b{ctr, ctr + 64, ctr + 1024, nb, mode};
      const int iters = 2000;
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(barrier_loop, dim3(nb), dim3(512), 0, st, b, iters, cyc);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("barrier mode %d blocks %3d: %.3f us per barrier\n", mode, nb, ms * 1e3 / iters);
    }
  // phased streaming
  const int layers = 28, nb = ncu < 256 ? ncu : 256;
  const size_t nw = (size_t)nb * 8;
  const size_t wbytes = (size_t)layers * CL * nw * 1024;
  u32x4_t* W;
  CK(hipMalloc(&W, wbytes));
  CK(hipMemset(W, 0x3c, wbytes));
  unsigned short* act;
  CK(hipMalloc(&act, 1 << 20));
  CK(hipMemset(act, 0, 1 << 20));
  float* sink;
  CK(hipMalloc(&sink, 64));
  printf("weights %.1f MB per layer, %.2f GB total\n", CL * nw * 1024 / 1e6, wbytes / 1e9);
  for (int mode = 0; mode < 2; ++mode)
    for (int pf = 0; pf < 2; ++pf)
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(ctr, 0, 4096 * 4, st));
        Bar b{ctr, ctr + 64, ctr + 1024, nb, mode};
        CK(hipEventRecord(e0, st));
        if (pf) hipLaunchKernelGGL(phased<1>, dim3(nb), dim3(512), 0, st, b, (const u32x4_t*)W, layers, act, sink);
        else    hipLaunchKernelGGL(phased<0>, dim3(nb), dim3(512), 0, st, b, (const u32x4_t*)W, layers, act, sink);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("phased barrier-mode %d prefetch %d: %.1f us per token-pass, %.2f us per layer, %.2f TB/s\n", mode, pf,
                        ms * 1e3, ms * 1e3 / layers, wbytes / (ms * 1e-3) / 1e12);
      }
  return 0;
}
