// Feasibility probe (not product code): can the weight stream of a batch-1 decode layer be overlapped with the
// all-to-all hand-offs between its five dependent stages WITHOUT a megakernel?
//
// Each stage is a 256-workgroup x 512-thread kernel (<= 128 VGPRs: two such workgroups fit a CU, so two stages are
// always co-resident and neither can starve the other of slots) that
//   1. issues its whole weight slice into registers (NLD x 16 B per thread),
//   2. waits in-kernel for its predecessor's arrival counter,
//   3. reads the predecessor's output vector, does the dot products, writes its own output,
//   4. bumps its own arrival counter.
// "launch" mode runs the stages on ONE stream (every hand-off is a kernel boundary, step 2 is skipped) - the structure
// of the shipped decode step.  "pdl" mode alternates the stages over 2 or 3 branches of one captured graph with no
// graph edge between them: stage j+1 is dispatched as soon as stage j-1 (same branch) has finished, i.e. while stage j
// still runs, so its weights stream during stage j's hand-off.  Stage sizes follow Qwen2-VL-2B:
// qkv 6 MB, attention (2 workgroups, 0.5 MB of K/V), o_proj 4 MB, gate/up 54 MB, down 26 MB per layer, x 28 layers.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 scripts/pdl_probe.cpp -o scripts/bin/pdl_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                 \
  do {                                                                                        \
    hipError_t e__ = (x);                                                                     \
    if (e__ != hipSuccess) {                                                                  \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
      exit(2);                                                                                \
    }                                                                                         \
  } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int WGS = 256, THREADS = 512;
enum { PROTO_NONE = 0, PROTO_FENCE = 1, PROTO_SC1 = 2 };

struct StageArgs {
  const u32x4* W;          // this stage's weights
  const uint16_t* x;       // predecessor's output (K bf16)
  uint16_t* y;             // this stage's output
  int K;                   // elements of x
  int nout;                // outputs per wave written
  unsigned* cnt_in;        // predecessor's arrival counter (nullptr: none)
  unsigned expect;         // arrivals to wait for
  unsigned* cnt_out;       // own arrival counter
  unsigned* err;           // set when a spin gives up
  float* stamps;           // [3] wall-clock stamps of workgroup 0 (us), or nullptr
  int proto;
  int iters;               // weight passes (1 = everything preloaded)
};

__device__ __forceinline__ float dot2(unsigned w, unsigned x, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w), __builtin_bit_cast(bf16x2, x), acc, false);
}

__device__ __forceinline__ void wait_counter(unsigned* cnt, unsigned expect, unsigned* err) {
  unsigned spins = 0;
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 22)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
  }
}

template <int NLD, int XR>   // XR: distinct x chunks per lane (K = 512 * XR elements when XR < NLD)
__global__ __launch_bounds__(THREADS, 4) void stage_kernel(StageArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // x as bf16 [K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long t0 = 0;
  if (a.stamps && blockIdx.x == 0 && tid == 0) t0 = wall_clock64();
  // ---- 1. the whole weight slice of this thread, in flight before anything else
  u32x4 w[NLD];
  const size_t stride = (size_t)WGS * THREADS;
  const u32x4* base = a.W + (size_t)blockIdx.x * THREADS + tid;
#pragma unroll
  for (int j = 0; j < NLD; ++j) w[j] = __builtin_nontemporal_load(base + (size_t)j * stride);
  __builtin_amdgcn_sched_barrier(0);
  // ---- 2. wait for the predecessor
  if (a.cnt_in && a.proto != PROTO_NONE) {
    if (tid == 0) {
      wait_counter(a.cnt_in, a.expect, a.err);
      if (a.proto == PROTO_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  unsigned long long t1 = 0;
  if (a.stamps && blockIdx.x == 0 && tid == 0) t1 = wall_clock64();
  // ---- 3. x -> LDS (8-byte agent-scope loads: L1 is bypassed, no acquire needed when the producer wrote through)
  {
    const unsigned long long* xs = reinterpret_cast<const unsigned long long*>(a.x);
    unsigned long long* xd = reinterpret_cast<unsigned long long*>(smem);
    const int n8 = a.K >> 2;
    for (int i = tid; i < n8; i += THREADS)
      xd[i] = a.proto == PROTO_SC1 ? __hip_atomic_load(xs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : xs[i];
  }
  __syncthreads();
  float acc = 0.f;
  const int nchunk = a.K >> 3;
  u32x4 xr[XR];
#pragma unroll
  for (int c = 0; c < XR; ++c) xr[c] = *reinterpret_cast<const u32x4*>(smem + (size_t)((lane + 64 * c) % nchunk) * 16);
  for (int it = 0; it < a.iters; ++it) {
    if (it > 0) {
#pragma unroll
      for (int j = 0; j < NLD; ++j) w[j] = __builtin_nontemporal_load(base + ((size_t)it * NLD + j) * stride);
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const u32x4 xv = xr[j % XR];
      acc = dot2(w[j][0], xv[0], acc);
      acc = dot2(w[j][1], xv[1], acc);
      acc = dot2(w[j][2], xv[2], acc);
      acc = dot2(w[j][3], xv[3], acc);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  // ---- outputs: `nout` values per wave (lane i writes value i), 4-byte agent-scope (write-through) stores
  if (lane < a.nout) {
    const int gw = blockIdx.x * (THREADS / 64) + wave;
    const float v = acc * 1e-3f + (float)lane;
    unsigned bits = __builtin_bit_cast(unsigned, v) >> 16;
    unsigned* yo = reinterpret_cast<unsigned*>(a.y) + ((size_t)gw * a.nout + lane) / 2;   // synthetic: pairs share a word
    if (a.proto == PROTO_SC1) __hip_atomic_store(yo, bits | (bits << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *yo = bits | (bits << 16);
  }
  // ---- 4. arrive
  if (a.proto != PROTO_NONE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (a.proto == PROTO_FENCE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(a.cnt_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (a.stamps && blockIdx.x == 0 && tid == 0) {
    const unsigned long long t2 = wall_clock64();
    a.stamps[0] = (float)(t0 % 100000000ull) * 0.01f;
    a.stamps[1] = (float)(t1 % 100000000ull) * 0.01f;
    a.stamps[2] = (float)(t2 % 100000000ull) * 0.01f;
  }
}

// attention stand-in: 2 workgroups, 256 KB of K/V each, a dependent arithmetic chain, 1536 outputs
__global__ __launch_bounds__(THREADS) void attn_kernel(StageArgs a) {
  const int tid = threadIdx.x;
  unsigned long long t0 = 0;
  if (a.stamps && blockIdx.x == 0 && tid == 0) t0 = wall_clock64();
  if (a.cnt_in && a.proto != PROTO_NONE) {
    if (tid == 0) {
      wait_counter(a.cnt_in, a.expect, a.err);
      if (a.proto == PROTO_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  unsigned long long t1 = 0;
  if (a.stamps && blockIdx.x == 0 && tid == 0) t1 = wall_clock64();
  u32x4 kv[32];
  const u32x4* base = a.W + (size_t)blockIdx.x * THREADS * 32 + tid;
#pragma unroll
  for (int j = 0; j < 32; ++j) kv[j] = base[(size_t)j * THREADS];
  float acc = (float)a.x[tid];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc = dot2(kv[j][0], kv[j][1], acc) + dot2(kv[j][2], kv[j][3], 0.f);
  for (int i = 0; i < 300; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);      // ~MFMA + softmax latency
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __shared__ float red[8];
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  const float s = red[0] + red[1] + red[2] + red[3] + red[4] + red[5] + red[6] + red[7];
  if (tid < 384) {   // 768 bf16 per workgroup
    unsigned bits = __builtin_bit_cast(unsigned, s * 1e-6f + (float)tid) >> 16;
    unsigned* yo = reinterpret_cast<unsigned*>(a.y) + blockIdx.x * 384 + tid;
    if (a.proto == PROTO_SC1) __hip_atomic_store(yo, bits | (bits << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *yo = bits | (bits << 16);
  }
  if (a.proto != PROTO_NONE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (a.proto == PROTO_FENCE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(a.cnt_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (a.stamps && blockIdx.x == 0 && tid == 0) {
    const unsigned long long t2 = wall_clock64();
    a.stamps[0] = (float)(t0 % 100000000ull) * 0.01f;
    a.stamps[1] = (float)(t1 % 100000000ull) * 0.01f;
    a.stamps[2] = (float)(t2 % 100000000ull) * 0.01f;
  }
}

__global__ void empty_kernel(unsigned* p) {
  if (p && threadIdx.x == 0 && blockIdx.x == 0xffffff) p[0] = 1;
}

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)i * 0x9E3779B1u + seed;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15;
    p[i] = (x & 0x807f807fu) | 0x3c003c00u;      // two small bf16 values of random sign
  }
}

struct StageDef { int nld; int K; int nout; bool attn; const char* name; };

static void launch_stage(const StageDef& d, const StageArgs& a, hipStream_t st) {
  if (d.attn) { hipLaunchKernelGGL(attn_kernel, dim3(2), dim3(THREADS), 0, st, a); return; }
  const size_t lds = (size_t)a.K * 2;
#define L(N, X) case N: hipLaunchKernelGGL((stage_kernel<N, X>), dim3(WGS), dim3(THREADS), lds, st, a); break
  switch (d.nld) {
    L(2, 2); L(3, 3); L(13, 13); L(26, 3);
    default: fprintf(stderr, "no stage_kernel<%d>\n", d.nld); exit(4);
  }
#undef L
}

int main(int argc, char** argv) {
  int layers = 28, reps = 200;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--layers") && i + 1 < argc) layers = atoi(argv[++i]);
    if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
  }
  const size_t unit = (size_t)WGS * THREADS * 16;     // bytes per preloaded 16-B load across the grid (2 MiB)
  const std::vector<StageDef> layer = {{3, 1536, 1, false, "qkv"}, {0, 1536, 0, true, "attn"}, {2, 1536, 1, false, "o_proj"},
                                       {26, 1536, 4, false, "gate_up"}, {13, 8960, 1, false, "down"}};
  size_t layer_bytes = 0;
  for (const StageDef& d : layer) layer_bytes += d.attn ? (size_t)2 * THREADS * 32 * 16 : (size_t)d.nld * unit;
  const size_t total = layer_bytes * layers;
  printf("synthetic layer: %.1f MB in %zu stages, %d layers = %.2f GB per step\n", layer_bytes / 1e6, layer.size(), layers, total / 1e9);
  char* W;
  CK(hipMalloc(&W, total));
  fill_kernel<<<4096, 256>>>((unsigned*)W, total / 4, 7);
  const int nstage = (int)layer.size() * layers;
  uint16_t* vec;                               // ping-pong activation vectors, 32 KB apart
  CK(hipMalloc(&vec, (size_t)(nstage + 1) * 32768));
  fill_kernel<<<256, 256>>>((unsigned*)vec, (size_t)(nstage + 1) * 32768 / 4, 9);
  unsigned* cnt;
  CK(hipMalloc(&cnt, (size_t)(nstage + 2) * 64 * 4));
  float* stamps;
  CK(hipMalloc(&stamps, (size_t)nstage * 4 * 4));
  CK(hipMemset(stamps, 0, (size_t)nstage * 4 * 4));
  CK(hipDeviceSynchronize());

  // ---- E1: kernel-boundary cost of a captured chain of empty kernels
  {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 140; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, (unsigned*)nullptr);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("E1 empty 256-WG kernels in a graph: %.2f us per kernel\n", ms * 1e3 / 50 / 140);
  }

  struct Mode { const char* name; int nstreams; int proto; };
  const Mode modes[] = {{"launch (1 stream, boundaries)", 1, PROTO_NONE}, {"pdl 2 branches, fences", 2, PROTO_FENCE},
                        {"pdl 2 branches, sc1", 2, PROTO_SC1}, {"pdl 3 branches, fences", 3, PROTO_FENCE},
                        {"pdl 3 branches, sc1", 3, PROTO_SC1}, {"launch again", 1, PROTO_NONE}};
  unsigned* err = cnt + (size_t)(nstage + 1) * 64;
  for (const Mode& md : modes) {
    hipStream_t s[3];
    for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(cnt, 0, (size_t)(nstage + 2) * 64 * 4, s[0]));
    hipEvent_t fork, joins[3];
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventRecord(fork, s[0]));
    for (int i = 1; i < md.nstreams; ++i) CK(hipStreamWaitEvent(s[i], fork, 0));
    size_t woff = 0;
    for (int j = 0; j < nstage; ++j) {
      const StageDef& d = layer[j % layer.size()];
      const StageDef& prev = layer[(j + layer.size() - 1) % layer.size()];
      StageArgs a{};
      a.W = (const u32x4*)(W + woff);
      woff += d.attn ? (size_t)2 * THREADS * 32 * 16 : (size_t)d.nld * unit;
      a.x = vec + (size_t)j * 16384;
      a.y = vec + (size_t)(j + 1) * 16384;
      a.K = d.K;
      a.nout = d.nout;
      a.cnt_in = j > 0 ? cnt + (size_t)(j - 1) * 64 : nullptr;
      a.expect = prev.attn ? 2u : (unsigned)WGS;
      a.cnt_out = cnt + (size_t)j * 64;
      a.err = err;
      a.stamps = stamps + (size_t)j * 4;
      a.proto = md.proto;
      a.iters = 1;
      launch_stage(d, a, s[j % md.nstreams]);
    }
    for (int i = 1; i < md.nstreams; ++i) {
      CK(hipEventCreateWithFlags(&joins[i], hipEventDisableTiming));
      CK(hipEventRecord(joins[i], s[i]));
      CK(hipStreamWaitEvent(s[0], joins[i], 0));
    }
    CK(hipStreamEndCapture(s[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s[0]));
    CK(hipStreamSynchronize(s[0]));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s[0]));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, s[0]));
    CK(hipEventRecord(e1, s[0]));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned herr = 0;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    const double us = ms * 1e3 / reps;
    printf("%-34s %8.1f us/step  %6.2f us/layer  %5.2f TB/s  spin-giveup=%u\n", md.name, us, us / layers, total / us * 1e-6, herr);
    // timeline of layer 10 (stage start / flag seen / end, us relative to the layer's first start)
    std::vector<float> hs((size_t)nstage * 4);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 4, hipMemcpyDeviceToHost));
    const int L0 = 10 * (int)layer.size();
    if (layers > 11) {
      const float ref = hs[(size_t)L0 * 4];
      printf("   layer 10:");
      for (int k = 0; k < (int)layer.size() + 1; ++k) {
        const float* p = &hs[(size_t)(L0 + k) * 4];
        printf("  %s[%.1f %.1f %.1f]", layer[k % layer.size()].name, p[0] - ref, p[1] - ref, p[2] - ref);
      }
      printf("\n");
    }
    fflush(stdout);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    for (int i = 0; i < 3; ++i) CK(hipStreamDestroy(s[i]));
  }
  return 0;
}
