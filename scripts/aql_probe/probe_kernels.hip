// Kernels of the AQL fence probe (scripts/aql_probe/aql_probe.cpp).  No blockDim / gridDim (those are hidden kernel arguments of a
// code-object-v5 kernel; the probe fills none): the workgroup size is the constant below.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define TPB 256

extern "C" __global__ __launch_bounds__(TPB) void k_empty() {}

// one link of a dependent chain: out[i] = in[i] + 1 over n floats, first workgroups only
extern "C" __global__ __launch_bounds__(TPB) void k_chain_plain(const float* in, float* out, int n) {
  const int i = __builtin_amdgcn_workgroup_id_x() * TPB + __builtin_amdgcn_workitem_id_x();
  if (i < n) out[i] = in[i] + 1.f;
}

// the same with agent-scope accesses (sc1 loads / stores on gfx950: they go past the XCD-private L2 state)
extern "C" __global__ __launch_bounds__(TPB) void k_chain_agent(const float* in, float* out, int n) {
  const int i = __builtin_amdgcn_workgroup_id_x() * TPB + __builtin_amdgcn_workitem_id_x();
  if (i < n) {
    const float v = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(out + i, v + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// a small GEMV-like link: every workgroup streams its slice of w (read-only, plain loads) against x (agent-scope loads) and writes
// one y per wave (agent-scope store); rows x cols bf16-sized words
extern "C" __global__ __launch_bounds__(TPB) void k_gemv_agent(const uint32_t* __restrict__ w, const float* x, float* y, int rows, int cols32) {
  const int wave = (__builtin_amdgcn_workgroup_id_x() * TPB + __builtin_amdgcn_workitem_id_x()) >> 6, lane = __builtin_amdgcn_workitem_id_x() & 63;
  if (wave >= rows) return;
  float acc = 0.f;
  for (int c = lane; c < cols32; c += 64) {
    const uint32_t q = w[(size_t)wave * cols32 + c];
    const float xv = __hip_atomic_load(x + (c & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc += __uint_as_float(q << 16) * xv + __uint_as_float(q & 0xffff0000u) * xv;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) __hip_atomic_store(y + wave, acc * 1e-3f + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------------------
// A realistic link: the decode step's row-wave GEMV in miniature (2 rows per wave, K = 1536 bf16 = 3 chunks of 64 lanes x 16 B,
// all weight loads issued up front).  y = W x over `rows` rows; x is the previous link's y.  flag protocol (WAIT = true): the
// weights do not depend on the previous link, so they are issued first; then one lane spins on the previous link's arrival
// counter (bounded), then x is read with sc1 loads, and after the sc1 stores every workgroup arrives on the counter.  With WAIT
// the packets carry no barrier bit: link i + 1 is dispatched while link i runs and its weight stream overlaps link i's tail.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct LinkArgs { const uint32_t* w; const uint32_t* x; uint32_t* y; int* flag; int rows; int wait_for; int* err; int link; int n_wg;
                  const uint32_t* pf; long long pf_n16; int pf_wgs; int pad; };

__device__ __forceinline__ float dot8(u32x4 w, u32x4 x, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc += __uint_as_float(w[i] << 16) * __uint_as_float(x[i] << 16);
    acc += __uint_as_float(w[i] & 0xffff0000u) * __uint_as_float(x[i] & 0xffff0000u);
  }
  return acc;
}

template <bool WAIT>
__device__ __forceinline__ void link_body(const LinkArgs& a) {
  const int tid = __builtin_amdgcn_workitem_id_x(), lane = tid & 63, wave = tid >> 6;
  if ((int)__builtin_amdgcn_workgroup_id_x() >= a.n_wg) {
    // prefetch workgroups (the launch carries pf_wgs of them behind its n_wg working ones): stream a LATER link's weights with plain
    // loads and drop them - the launch is latency-bound and leaves most of the HBM bandwidth idle; what is read here sits in the
    // memory-side cache (256 MB) when its own launch asks for it
    const int p = __builtin_amdgcn_workgroup_id_x() - a.n_wg;
    const long long per = (a.pf_n16 + a.pf_wgs - 1) / a.pf_wgs, lo = p * per, hi = min(lo + per, a.pf_n16);
    const u32x4* src = reinterpret_cast<const u32x4*>(a.pf);
    u32x4 acc = {0, 0, 0, 0};
    for (long long i = lo + tid; i < hi; i += TPB * 4) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[min(i + (long long)TPB * u, hi - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u && a.rows < 0) a.y[0] = 1;      // never: keeps the loads
    return;
  }
  const int gw = __builtin_amdgcn_workgroup_id_x() * 4 + wave;
  const int r0 = min(gw * 2, a.rows - 1), r1 = min(gw * 2 + 1, a.rows - 1);
  u32x4 wv[2][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    wv[0][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.w + (size_t)r0 * 768) + lane + 64 * c);
    wv[1][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.w + (size_t)r1 * 768) + lane + 64 * c);
  }
  if (WAIT) {
    // the previous link's LAST workgroup stores its link number into the link's own word (a.flag[16 + 32 * (link & 63)]: one
    // 128-byte line per link in flight, written once); polling is a relaxed sc1 load - an acquire in the loop would invalidate
    // the L2 once per iteration per workgroup
    if (tid == 0 && a.wait_for >= 0) {
      const int* f = a.flag + 32 + 32 * (a.wait_for & 63);
      int it = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.wait_for) {
        if (++it > (1 << 18)) { *a.err = 1; break; }        // bounded: a broken protocol shows up as a flag, not as a hung GPU
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
  }
  u32x4 xv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const u32x4* p = reinterpret_cast<const u32x4*>(a.x) + lane + 64 * c;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(xv[c]) : "v"(p) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) { acc0 = dot8(wv[0][c], xv[c], acc0); acc1 = dot8(wv[1][c], xv[c], acc1); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { acc0 += __shfl_xor(acc0, o, 64); acc1 += __shfl_xor(acc1, o, 64); }
  if (lane == 0 && gw * 2 + 1 < a.rows) {
    // keep the values bounded and the dependency real: two bf16 in one word
    const uint32_t o = (__float_as_uint(acc0 * 1e-3f + 0.5f) >> 16) | (__float_as_uint(acc1 * 1e-3f + 0.5f) & 0xffff0000u);
    __hip_atomic_store(a.y + (gw % 768), o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (WAIT) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's sc1 store has been acknowledged
    __syncthreads();
    if (tid == 0) {
      // arrival counter of THIS link (its own line); the last arriver publishes the link number for the next link's pollers
      int* cnt = a.flag + 32 + 32 * 64 + 32 * (a.link & 63);
      const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == a.n_wg - 1) {
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.flag + 32 + 32 * (a.link & 63), a.link, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
extern "C" __global__ __launch_bounds__(TPB) void k_link_barrier(LinkArgs a) { link_body<false>(a); }
extern "C" __global__ __launch_bounds__(TPB) void k_link_flag(LinkArgs a) { link_body<true>(a); }
