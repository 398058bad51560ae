// Weight / KV prefetch for the decode step on gfx950 (library-internal, see csrc/internal.h).
//
// A batch-1 decode step is a chain of ~140 dependent weight-streaming kernels (reference: the Python layer loop of
// Qwen2Model, mlx_vlm/models/qwen2_vl/language.py:170-200, under generate_step, generate/ar.py:334-389).  Each link
// pays a kernel boundary + a first-touch round trip to HBM (page-table walk included) before its stream starts, and
// the attention link moves almost no bytes, so the HBM is idle for more than half of every layer
// (profiles/r01_bench_kernel_stats_v8.txt: 34.9 us per layer for 93.6 MB).  The 256 MB Infinity Cache holds two
// layers' weights: while layer i runs, a side kernel on a second graph branch pulls layer i + 1 (weights and the K/V
// pages its attention will read) through the memory-side cache with plain loads whose data is discarded.  The chain
// kernels then find their first lines - and most of their stream - on die.  No data is exchanged with the chain:
// the side kernel only reads, so there is nothing to order and no result depends on it.
//
// Two forms:
//   * one launch per layer on the side branch, started by a graph edge from the chain (event-paced);
//   * one persistent launch per step that walks a device-resident list and starts item k when the chain's pacing
//     word reaches item[k].need (flag-paced: no graph edge leaves the chain).
#include "common.cuh"
#include "internal.h"

namespace {

constexpr int PF_THREADS = 256;
constexpr int PF_UNROLL = 8;
constexpr size_t PF_ROW = (size_t)PF_THREADS * 16;          // bytes one workgroup covers per load instruction
constexpr size_t PF_CHUNK = PF_ROW * PF_UNROLL;             // 32 KiB per workgroup iteration

// touch [base, base + bytes) once: chunk c belongs to workgroup (c + rot) % nwg
__device__ __forceinline__ void pf_range(const char* __restrict__ base, size_t bytes, int wg, int nwg, int rot,
                                         unsigned& sink) {
  if (bytes < 16) return;
  const size_t nchunk = (bytes + PF_CHUNK - 1) / PF_CHUNK;
  const size_t last = (bytes - 16) & ~(size_t)15;
  size_t c = (size_t)((wg + nwg - (rot % nwg)) % nwg);
  for (; c < nchunk; c += (size_t)nwg) {
    u32x4_t v[PF_UNROLL];
#pragma unroll
    for (int j = 0; j < PF_UNROLL; ++j) {
      size_t off = c * PF_CHUNK + (size_t)j * PF_ROW + (size_t)threadIdx.x * 16;
      off = off < last ? off : last;                         // branch-free tail: re-touch the last line
      v[j] = *reinterpret_cast<const u32x4_t*>(base + off);
    }
#pragma unroll
    for (int j = 0; j < PF_UNROLL; ++j) sink ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
  }
}

__device__ __forceinline__ void pf_item(const VlmPfItem& it, const VlmPfKv& kv, int wg, int nwg, unsigned& sink) {
  int rot = 0;
  for (int s = 0; s < it.nseg; ++s) {
    pf_range(static_cast<const char*>(it.seg[s].p), it.seg[s].bytes, wg, nwg, rot, sink);
    rot += (int)((it.seg[s].bytes + PF_CHUNK - 1) / PF_CHUNK);
  }
  if (it.kbase && kv.ctx) {
    // the pages the layer's decode attention will read: ceil((ctx + 1) / 64) per sequence, K and V pools
    for (int b = 0; b < kv.B; ++b) {
      const int np = min((kv.ctx[b] + 1 + 63) >> 6, kv.max_pages);
      for (int p = wg; p < 2 * np; p += nwg) {
        const int pg = p >> 1;
        const size_t page = kv.block_table ? (size_t)kv.block_table[(size_t)b * kv.max_pages + pg]
                                           : (size_t)b * kv.max_pages + pg;
        const char* src = static_cast<const char*>((p & 1) ? it.vbase : it.kbase) + page * kv.page_bytes;
        pf_range(src, kv.page_bytes, 0, 1, 0, sink);
      }
    }
  }
}

__global__ __launch_bounds__(PF_THREADS) void prefetch_kernel(VlmPfItem it, VlmPfKv kv, unsigned* __restrict__ never) {
  unsigned sink = 0;
  pf_item(it, kv, blockIdx.x, gridDim.x, sink);
  if (sink == 0x9e3779b9u && never) never[0] = sink;        // keeps the loads alive; `never` is nullptr
}

__global__ __launch_bounds__(PF_THREADS) void prefetch_persistent_kernel(const VlmPfItem* __restrict__ items, int n_items,
                                                                         VlmPfKv kv, int* progress, unsigned* exit_count,
                                                                         unsigned* __restrict__ never) {
  __shared__ int s_prog;
  unsigned sink = 0;
  for (int k = 0; k < n_items; ++k) {
    const VlmPfItem it = items[k];
    if (threadIdx.x == 0) {
      int p = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (p < it.need && spins < (1u << 22)) {            // bounded: a lost pacing word ends in a skipped prefetch
        __builtin_amdgcn_s_sleep(8);
        p = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++spins;
      }
      s_prog = p;
    }
    __syncthreads();
    const int p = s_prog;
    __syncthreads();
    if (p >= it.need + 2) continue;                           // the chain is already past this item
    pf_item(it, kv, blockIdx.x, gridDim.x, sink);
  }
  // last workgroup out re-arms the pacing word for the next step (nobody polls it any more)
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(exit_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      __hip_atomic_store(exit_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(progress, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (sink == 0x9e3779b9u && never) never[0] = sink;
}

}  // namespace

int vlm_prefetch_launch(const VlmPfItem* item, const VlmPfKv* kv, int wgs, void* stream) {
  if (!item || !kv || wgs <= 0) return VLM_ERR_ARG;
  hipLaunchKernelGGL(prefetch_kernel, dim3(wgs), dim3(PF_THREADS), 0, (hipStream_t)stream, *item, *kv, (unsigned*)nullptr);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

int vlm_prefetch_persistent_launch(const VlmPfItem* items_dev, int n_items, const VlmPfKv* kv, int* progress,
                                   unsigned* exit_count, int wgs, void* stream) {
  if (!items_dev || !kv || !progress || !exit_count || wgs <= 0 || n_items <= 0) return VLM_ERR_ARG;
  hipLaunchKernelGGL(prefetch_persistent_kernel, dim3(wgs), dim3(PF_THREADS), 0, (hipStream_t)stream, items_dev, n_items, *kv,
                     progress, exit_count, (unsigned*)nullptr);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
