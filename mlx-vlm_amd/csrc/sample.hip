// Sampler kernels for gfx950 over the vocabulary row (V = 151,936 for Qwen2-VL):
//   logprobs = logits - logsumexp(logits)          (reference mlx_vlm/generate/ar.py:368)
//   greedy   = argmax(logprobs), lowest index wins  (reference mlx_vlm/sample_utils.py:63-64)
//   the filters of make_sampler in its order         (sample_utils.py:10-89: top-n-sigma 181-212, p-less 215-236,
//                                                     typical-p 321-345, top-p 289-318, min-p 239-286, xtc 348-376, top-k 169-175)
//   categorical(logprobs / temp) by Gumbel-max      (sample_utils.py:385-387)
// The row is 300 KB: everything is a couple of passes of 16-byte loads with
// wavefront-shuffle reductions; logprobs are produced in the logits dtype (bf16)
// exactly as the reference does (lse rounded to bf16, then the difference rounded),
// and the filters follow MLX's TYPED graph over bf16 (every elementary op rounds; python
// scalars are converted to bf16 first): pinned by tests/golden/samplers_ref.npz.
//
// Because logprobs are bf16 there are only 65,536 distinct keys: top-k, top-p and
// min_tokens_to_keep are done EXACTLY with a histogram over the keys (the half that
// holds values <= 0 in LDS) instead of a sort; typical-p, whose order is not the
// log-prob's, sorts the indices with a stable in-kernel radix sort.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "internal.h"
#include "../../include/vlm_hip.h"

namespace {

constexpr int NBLK = 64;  // blocks per vocabulary row
// Workspace row: every per-row array of the workspace lives in ONE block of ROW_W 4-byte words per batch row, at the same offset
// for every call - [NBLK][2] f32 lse partials | [NBLK] f32 cand_v | [NBLK] i32 cand_i | [65536] u32 key histogram | [CTL_WORDS] u32
// control words of the split top-p path.  Row b of array X is X_base + b * ROW_W whatever the call's B: a state whose workspace
// was sized for 8 rows and that steps at 8, then 4, then 8 rows finds the persistent arrays (histogram and control words, zero /
// re-armed between calls) of row b in the same words every time.  (Rounds 4-5 laid the arrays out [B][...] back to back: a
// narrower step then found another step's Gumbel candidates where its histogram should be zero - ADVICE r05.)
constexpr int CTL_WORDS = 128;
constexpr int CTL_ROW_TICKET = 120;   // control word of a row's block: arrivals of the row's blocks in the fused greedy tail (zero between launches)
constexpr int ROW_LSE = 0, ROW_CV = 2 * NBLK, ROW_CI = 3 * NBLK, ROW_HIST = 4 * NBLK, ROW_CTL = ROW_HIST + 65536;
constexpr int ROW_W = ROW_CTL + CTL_WORDS;
static_assert(ROW_HIST % 4 == 0 && ROW_W % 4 == 0, "the histogram of every row is cleared with 16-byte stores");

// order-preserving map bf16 bits -> uint16 (ascending)
__device__ __forceinline__ uint32_t bf_key(bf16_t b) { return (b & 0x8000u) ? (uint32_t)(~b & 0xffffu) : (uint32_t)(b | 0x8000u); }
__device__ __forceinline__ bf16_t key_bf(uint32_t k) { return (k & 0x8000u) ? (bf16_t)(k & 0x7fffu) : (bf16_t)(~k & 0xffffu); }

__global__ __launch_bounds__(256) void lse_partial_kernel(const bf16_t* __restrict__ logits, int ld, int V,
                                                          float* __restrict__ ws) {
  __shared__ float red[16];
  const int b = blockIdx.y, blk = blockIdx.x;
  const int per = ((V + NBLK - 1) / NBLK + 7) & ~7;
  const int lo = blk * per, hi = min(V, lo + per);
  const bf16_t* row = logits + (size_t)b * ld;
  float m = -INFINITY;
  for (int i = lo + threadIdx.x; i < hi; i += 256) m = fmaxf(m, bf2f(row[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  if (m > -INFINITY)
    for (int i = lo + threadIdx.x; i < hi; i += 256) s += expf(bf2f(row[i]) - m);
  s = block_sum(s, red + 4);
  if (threadIdx.x == 0) { ws[(size_t)b * ROW_W + blk * 2] = m; ws[(size_t)b * ROW_W + blk * 2 + 1] = s; }
}

// logprobs (bf16) + per-block argmax candidate
__global__ __launch_bounds__(256) void logprob_argmax_kernel(const bf16_t* __restrict__ logits, int ld, int V,
                                                             const float* __restrict__ ws, bf16_t* __restrict__ logprobs,
                                                             int ldlp, float* __restrict__ cand_v, int* __restrict__ cand_i) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int b = blockIdx.y, blk = blockIdx.x;
  // merge the NBLK (= 64 = one wavefront) partials: lane i takes partial i, shuffle reductions - every wave
  // does it redundantly (a serial loop of 128 dependent L2 loads cost ~10 us here)
  const int li = threadIdx.x & 63;
  const float pm = ws[(size_t)b * ROW_W + li * 2], ps = ws[(size_t)b * ROW_W + li * 2 + 1];
  const float m = wave_max(pm);
  const float s = wave_sum(pm > -INFINITY ? ps * expf(pm - m) : 0.f);
  const float lse = rbf(m + logf(s));          // logsumexp materialised in the logits dtype
  const int per = ((V + NBLK - 1) / NBLK + 7) & ~7;
  const int lo = blk * per, hi = min(V, lo + per);
  const bf16_t* row = logits + (size_t)b * ld;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const bf16_t lpb = f2bf(bf2f(row[i]) - lse);
    if (logprobs) logprobs[(size_t)b * ldlp + i] = lpb;
    const float lp = bf2f(lpb);
    if (lp > best || (lp == best && i < besti)) { best = lp; besti = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < besti)) { best = sv[w]; besti = si[w]; }
    cand_v[(size_t)b * ROW_W + blk] = best;
    cand_i[(size_t)b * ROW_W + blk] = besti;
  }
}

__global__ __launch_bounds__(64) void argmax_final_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i,
                                                          int* __restrict__ tok) {
  const int b = blockIdx.x;
  float best = cand_v[(size_t)b * ROW_W + threadIdx.x];
  int besti = cand_i[(size_t)b * ROW_W + threadIdx.x];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if (threadIdx.x == 0) tok[b] = besti == 0x7fffffff ? 0 : besti;   // (sampling over a row with every token removed: a valid index anyway)
}

// The SAMPLED step's tail in one launch (the greedy step has logprob_argmax_tail_kernel): the pick of argmax_final_kernel for every
// row, then vlm_decode_advance's bookkeeping (cache.py:362 offset += 1, language.py:476-509 pos, token ring, step counter) and the
// embedding gather of the NEXT step (nn.Embedding, language.py:164,179) - the two launches a sampled step used to spend around
// the sampler.  One workgroup; B <= TAIL_MAX_B.
struct SampleTail {
  int *ctx, *pos, *out_ring, *step;
  int ring_len;
  const bf16_t* embed;
  bf16_t* h;
  int D, ldh;
};

// the next step's input rows h[r] = embed[tok[r]] by ONE workgroup (the last block of a fused tail): 8 loads of a thread in flight
// before the first store - one dependent HBM round trip per 16-byte piece made the 16-row greedy tail 27-32 us long (12 trips)
__device__ __forceinline__ void gather_rows_256(const bf16_t* __restrict__ embed, bf16_t* __restrict__ h, const int* s_tok, int B,
                                                int D, int ldh, int tid) {
  const int cpr = D >> 3, total = B * cpr;
  for (int i0 = tid; i0 < total; i0 += 256 * 8) {
    u32x4_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + 256 * u, total - 1), r = i / cpr, c = i % cpr;
      v[u] = reinterpret_cast<const u32x4_t*>(embed + (size_t)s_tok[r] * D)[c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u;
      if (i < total) reinterpret_cast<u32x4_t*>(h + (size_t)(i / cpr) * ldh)[i % cpr] = v[u];
    }
  }
}

__global__ __launch_bounds__(256) void argmax_final_advance_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i,
                                                                   int* __restrict__ tok, int B, int V, SampleTail t) {
  __shared__ int s_tok[64];
  const int tid = threadIdx.x, li = tid & 63, wave = tid >> 6;
  for (int r = wave; r < B; r += 4) {
    float bv = cand_v[(size_t)r * ROW_W + li];
    int bi = cand_i[(size_t)r * ROW_W + li];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (li == 0) s_tok[r] = (unsigned)bi >= (unsigned)V ? 0 : bi;      // (every token removed / no candidate: a valid index anyway)
  }
  __syncthreads();
  const int st = *t.step;
  if (tid < B) {
    const int y = s_tok[tid];
    tok[tid] = y;
    t.ctx[tid] += 1;
    t.pos[tid] += 1;
    if (t.out_ring) t.out_ring[(size_t)(st % t.ring_len) * B + tid] = y;
  }
  gather_rows_256(t.embed, t.h, s_tok, B, t.D, t.ldh, tid);
  __syncthreads();
  if (tid == 0) *t.step = st + 1;
}


// ------------------------------------------------------------------------------------------
// Greedy tail of a decode step in ONE launch: logprobs + per-block argmax candidates as logprob_argmax_kernel, then the
// LAST block to arrive (agent-scope ticket) reduces the candidates, writes the token and does what used to be three more
// launches: the bookkeeping of vlm_decode_advance (cache.py:362 offset += 1, language.py:476-509 pos = offset + delta,
// token ring, step counter) and the embedding gather of the NEXT step (nn.Embedding, language.py:164,179) into h.
// Hand-off recipe (cdna_hip_programming.md, Guideline 16): plain candidate stores -> release fence (agent) -> asm
// vmcnt(0) -> relaxed agent fetch_add; the last arriver: acquire fence (agent) -> barrier -> plain loads.
// ------------------------------------------------------------------------------------------
constexpr int TAIL_MAX_B = 64;

__global__ __launch_bounds__(256) void logprob_argmax_tail_kernel(
    const bf16_t* __restrict__ logits, int ld, int V, const float* __restrict__ ws, bf16_t* __restrict__ logprobs, int ldlp,
    float* cand_v, int* cand_i, unsigned* ticket, int* __restrict__ tok, int* __restrict__ ctx, int* __restrict__ pos,
    int* __restrict__ out_ring, int ring_len, int* __restrict__ step, const bf16_t* __restrict__ embed,
    bf16_t* __restrict__ h, int D, int ldh, int B) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ int s_last;
  __shared__ int s_tok[TAIL_MAX_B];
  const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int li = tid & 63;
  const float pm = ws[(size_t)b * ROW_W + li * 2], ps = ws[(size_t)b * ROW_W + li * 2 + 1];
  const float m = wave_max(pm);
  const float s = wave_sum(pm > -INFINITY ? ps * expf(pm - m) : 0.f);
  const float lse = rbf(m + logf(s));          // logsumexp materialised in the logits dtype (ar.py:368)
  const int per = ((V + NBLK - 1) / NBLK + 7) & ~7;
  const int lo = blk * per, hi = min(V, lo + per);
  const bf16_t* row = logits + (size_t)b * ld;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  // 8 elements per thread and trip where rows and pitches allow 16-byte accesses (elementwise + an order-free argmax with the
  // index tie rule: the same log-probs and candidate as the scalar loop); the ragged end of the last block stays scalar
  const bool vec = ((ld | ldlp) & 7) == 0 && (((uintptr_t)logits | (uintptr_t)logprobs) & 15) == 0;
  const int hi8 = vec ? lo + ((hi - lo) & ~7) : lo;
  for (int i = lo + tid * 8; i < hi8; i += 256 * 8) {
    const u32x4_t w = *reinterpret_cast<const u32x4_t*>(row + i);
    u32x4_t o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bf16_t l0 = f2bf(bf_lo(w[q]) - lse), l1 = f2bf(bf_hi(w[q]) - lse);
      o[q] = (uint32_t)l0 | ((uint32_t)l1 << 16);
      const float p0 = bf2f(l0), p1 = bf2f(l1);
      if (p0 > best || (p0 == best && i + 2 * q < besti)) { best = p0; besti = i + 2 * q; }
      if (p1 > best || (p1 == best && i + 2 * q + 1 < besti)) { best = p1; besti = i + 2 * q + 1; }
    }
    if (logprobs) *reinterpret_cast<u32x4_t*>(logprobs + (size_t)b * ldlp + i) = o;
  }
  for (int i = hi8 + tid; i < hi; i += 256) {
    const bf16_t lpb = f2bf(bf2f(row[i]) - lse);
    if (logprobs) logprobs[(size_t)b * ldlp + i] = lpb;
    const float lp = bf2f(lpb);
    if (lp > best || (lp == best && i < besti)) { best = lp; besti = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < besti)) { best = sv[w]; besti = si[w]; }
    // publish: the candidates as agent-scope atomic stores (written through to where every XCD reads them), their completion,
    // then the ticket.  (Rounds 2-5 used plain stores + an agent-scope RELEASE fence: on this part that fence writes the XCD's
    // L2 back - every dirty line in it, i.e. the log-probs all its blocks are storing - once per block; 1024 of them made the
    // 16-row launch 32 us long, round 6.  The last block reads the candidates with agent-scope atomic loads, as before.)
    __hip_atomic_store(cand_v + (size_t)b * ROW_W + blk, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(cand_i + (size_t)b * ROW_W + blk, besti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Two levels (round 6): the row's own ticket (a control word of the row's workspace block, 260 KB from its neighbours'),
    // then the last block of each row on the launch's ticket.  One ticket for all NBLK x B blocks serialises 1024 agent-scope
    // fetch_adds on ONE address at a 16-row step: the launch took 32 us, against 8.8 at one row (64 arrivals).
    unsigned* row_ticket = reinterpret_cast<unsigned*>(const_cast<float*>(ws)) + (size_t)b * ROW_W + ROW_CTL + CTL_ROW_TICKET;
    int last = 0;
    if (__hip_atomic_fetch_add(row_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
      __hip_atomic_store(row_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // every block of the row has arrived: re-arm
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.y - 1u;
    }
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- last block: final argmax per row (64 candidates = one wavefront), lowest index on ties (sample_utils.py:63-64)
  const int wave = tid >> 6;
  for (int r = wave; r < B; r += 4) {
    float bv = __hip_atomic_load(cand_v + (size_t)r * ROW_W + li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int bi = __hip_atomic_load(cand_i + (size_t)r * ROW_W + li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (li == 0) {
      // a row with no candidate (every logit NaN: `lp > best` never holds and the index stays 0x7fffffff) must not become an
      // address: the embedding gather below would read 6.6 TB past the table - a GPU memory fault instead of a wrong token.
      // Token 0 as argmax_final_kernel gives, and the event is counted in the word behind the ticket (ops.bad_argmax_rows)
      if ((unsigned)bi >= (unsigned)V) {
        bi = 0;
        __hip_atomic_fetch_add(ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_tok[r] = bi;
    }
  }
  __syncthreads();
  const int st = *step;
  if (tid < B) {
    const int t = s_tok[tid];
    tok[tid] = t;
    ctx[tid] += 1;
    pos[tid] += 1;
    if (out_ring) out_ring[(size_t)(st % ring_len) * B + tid] = t;
  }
  // next step's input embeddings
  gather_rows_256(embed, h, s_tok, B, D, ldh, tid);
  __syncthreads();
  if (tid == 0) {
    *step = st + 1;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every block has arrived: re-arm
  }
}


// ------------------------------------------------------------------------------------------
// logits processors (penalties / bias) - see include/vlm_hip.h::vlm_apply_logit_penalties
// ------------------------------------------------------------------------------------------
struct PenaltyK {
  int* hist;
  int* hist_len;
  int hist_cap;
  float rep_p;
  int rep_ctx;
  float pres_p;
  int pres_ctx;
  float freq_p;
  int freq_ctx;
  const int* bias_idx;
  const float* bias_val;
  int n_bias;
  const float* row_params;   // per-row form (continuous batch): [B][8] = rep_p, rep_ctx, pres_p, pres_ctx, freq_p, freq_ctx, n_bias, -
  int bias_stride;           // per-row form: row b's bias entries at bias_idx / bias_val + b * bias_stride
};

__global__ __launch_bounds__(1024) void logit_penalties_kernel(bf16_t* __restrict__ logits, int ld, int V,
                                                                const int* __restrict__ push_tok, PenaltyK p) {
  __shared__ int win[1024];
  __shared__ int s_len;
  const int b = blockIdx.x, tid = threadIdx.x;
  bf16_t* row = logits + (size_t)b * ld;
  int* hist = p.hist + (size_t)b * p.hist_cap;
  if (tid == 0) {
    int len = p.hist_len[b];
    if (push_tok) {
      hist[len % p.hist_cap] = push_tok[b];
      ++len;
      p.hist_len[b] = len;
    }
    s_len = len;
  }
  __syncthreads();
  const int len = s_len, n = min(len, p.hist_cap);
  // per-request processors of a continuous batch (ar.py:2584-2606 insert(..., logits_processors=)): row b reads its own
  // penalties / contexts / bias list from a device table the host rewrites as requests come and go (the captured step
  // keeps its arguments); a row with all-zero parameters passes through untouched
  if (p.row_params) {
    const float* rp = p.row_params + (size_t)b * 8;
    p.rep_p = rp[0]; p.rep_ctx = min((int)rp[1], p.hist_cap);
    p.pres_p = rp[2]; p.pres_ctx = min((int)rp[3], p.hist_cap);
    p.freq_p = rp[4]; p.freq_ctx = min((int)rp[5], p.hist_cap);
    p.n_bias = min((int)rp[6], p.bias_stride);
    p.bias_idx += (size_t)b * p.bias_stride;
    p.bias_val += (size_t)b * p.bias_stride;
  }
  // logit_bias: x + bf16(v), distinct indices
  for (int i = tid; i < p.n_bias; i += blockDim.x) {
    const int t = p.bias_idx[i];
    if (t >= 0 && t < V) row[t] = f2bf(bf2f(row[t]) + rbf(p.bias_val[i]));
  }
  __syncthreads();
  // one pass per processor: thread i owns entry i of the window (chronological order); the FIRST occurrence of a token
  // applies the update (once, or once per occurrence for the frequency penalty)
  auto pass = [&](int ctx, float pen, int kind) {
    const int w = min(ctx, n);
    if (w <= 0 || pen == 0.f) return;
    if (tid < w) win[tid] = hist[(len - w + tid) % p.hist_cap];
    __syncthreads();
    if (tid < w) {
      const int t = win[tid];
      bool first = true;
      int count = 0;
      for (int j = 0; j < w; ++j) {
        const bool same = win[j] == t;
        first = first && !(same && j < tid);
        count += same;
      }
      if (first && t >= 0 && t < V) {
        float x = bf2f(row[t]);
        const float pb = rbf(pen);                       // weak-typed python scalar: rounded to the logits dtype first
        if (kind == 0) x = x < 0.f ? rbf(x * pb) : rbf(x / pb);
        else if (kind == 1) x = rbf(x - pb);
        else for (int c = 0; c < count; ++c) x = rbf(x - pb);
        row[t] = f2bf(x);
      }
    }
    __syncthreads();
  };
  pass(p.rep_ctx, p.rep_p, 0);
  pass(p.pres_ctx, p.pres_p, 1);
  pass(p.freq_ctx, p.freq_p, 2);
}

// ------------------------------------------------------------------------------------------
// General sampler: one 1024-thread workgroup per row (the row is L2 resident).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float hash_uniform(uint32_t seed, uint32_t step, uint32_t row, uint32_t idx) {
  uint32_t x = seed ^ 0x9E3779B9u;
  x += (step + 1u) * 0x85EBCA6Bu;
  x ^= (row + 1u) * 0xC2B2AE35u;
  x += idx * 0x27D4EB2Fu;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// exclusive prefix over the 1024 threads of the block (thread order); `total` = block sum
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T* lds /* >= 17 */, T* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const T n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
  __syncthreads();
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T run = 0;
    for (int w = 0; w < 16; ++w) { const T t = lds[w]; lds[w] = run; run += t; }
    lds[16] = run;
  }
  __syncthreads();
  *total = lds[16];
  return inc - v + lds[wave];
}

constexpr bf16_t NEG_INF_BF = 0xff80u;

constexpr uint32_t KEY_NEG_INF = 0x007fu;   // bf_key(-inf)

// Among the elements whose key == tk (in index order) keep ranks [keep_lo, keep_hi), mask the rest.  Wave w owns a contiguous
// range of the row; an element's rank = matches in the waves before + in this wave's earlier iterations + in the lower lanes
// of this one (ballots).  Aligned rows: a lane takes 8 consecutive elements per iteration (one 16-byte load, 19 iterations for
// V = 151,936) and the lanes' match counts (0..8: four bits) are prefixed with four ballots; otherwise one element per lane.
// (History, profiles/r04_sampler_*: a contiguous chunk per THREAD cost 150 us of a top-p call - 64 cache lines per load;
// one element per lane with a ballot behind every load 59 us - 149 exposed load latencies per pass.)
__device__ __forceinline__ void mask_equal_by_rank(bf16_t* lp, int V, bool vec, uint32_t tk, uint32_t keep_lo, uint32_t keep_hi,
                                                   uint32_t* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t cnt = 0, rank = 0;
  if (vec && (V & 7) == 0) {
    const uint32_t tb = key_bf(tk);
    const int per = (((V + 15) >> 4) + 511) & ~511;
    const int w_lo = min(V, wave * per), w_hi = min(V, w_lo + per);
    auto hits = [&](const u32x4_t& w) -> uint32_t {
      const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
      return (uint32_t)((w0 & 0xffffu) == tb) | (uint32_t)((w0 >> 16) == tb) << 1 | (uint32_t)((w1 & 0xffffu) == tb) << 2 |
             (uint32_t)((w1 >> 16) == tb) << 3 | (uint32_t)((w2 & 0xffffu) == tb) << 4 | (uint32_t)((w2 >> 16) == tb) << 5 |
             (uint32_t)((w3 & 0xffffu) == tb) << 6 | (uint32_t)((w3 >> 16) == tb) << 7;
    };
    for (int i0 = w_lo; i0 < w_hi; i0 += 512) {
      const int base = i0 + lane * 8;
      if (base < w_hi) cnt += (uint32_t)__popc(hits(*reinterpret_cast<const u32x4_t*>(lp + base)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    __syncthreads();
    if (lane == 0) lds[wave] = cnt;
    __syncthreads();
    for (int w = 0; w < wave; ++w) rank += lds[w];
    for (int i0 = w_lo; i0 < w_hi; i0 += 512) {
      const int base = i0 + lane * 8;
      u32x4_t w = {0u, 0u, 0u, 0u};
      uint32_t m = 0;
      if (base < w_hi) { w = *reinterpret_cast<const u32x4_t*>(lp + base); m = hits(w); }
      const uint32_t n = (uint32_t)__popc(m);
      uint32_t before = 0, tot = 0;
#pragma unroll
      for (int bit = 0; bit < 4; ++bit) {
        const unsigned long long bb = __ballot((n >> bit) & 1u);
        before += (uint32_t)__popcll(bb & lt) << bit;
        tot += (uint32_t)__popcll(bb) << bit;
      }
      if (m) {
        uint32_t r = rank + before, wv[4] = {w[0], w[1], w[2], w[3]};
        bool changed = false;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if ((m >> e) & 1u) {
            if (r < keep_lo || r >= keep_hi) {
              wv[e >> 1] = (e & 1) ? (wv[e >> 1] & 0x0000ffffu) | ((uint32_t)NEG_INF_BF << 16) : (wv[e >> 1] & 0xffff0000u) | NEG_INF_BF;
              changed = true;
            }
            ++r;
          }
        if (changed) { u32x4_t o; o[0] = wv[0]; o[1] = wv[1]; o[2] = wv[2]; o[3] = wv[3]; *reinterpret_cast<u32x4_t*>(lp + base) = o; }
      }
      rank += tot;
    }
  } else {
    const int per = (((V + 15) >> 4) + 63) & ~63;
    const int w_lo = min(V, wave * per), w_hi = min(V, w_lo + per);
    for (int i0 = w_lo; i0 < w_hi; i0 += 64) {
      const int i = i0 + lane;
      cnt += (uint32_t)__popcll(__ballot(i < w_hi && bf_key(lp[min(i, V - 1)]) == tk));
    }
    __syncthreads();
    if (lane == 0) lds[wave] = cnt;
    __syncthreads();
    for (int w = 0; w < wave; ++w) rank += lds[w];
    for (int i0 = w_lo; i0 < w_hi; i0 += 64) {
      const int i = i0 + lane;
      const bool hit = i < w_hi && bf_key(lp[min(i, V - 1)]) == tk;
      const unsigned long long m = __ballot(hit);
      if (hit) {
        const uint32_t r = rank + (uint32_t)__popcll(m & lt);
        if (r < keep_lo || r >= keep_hi) lp[i] = NEG_INF_BF;
      }
      rank += (uint32_t)__popcll(m);
    }
  }
  __syncthreads();
}

// Row passes of the one-workgroup-per-row sampler: 8 elements (one 16-byte load) per lane per iteration when the row is 16-byte
// aligned (the engine's rows and every vocabulary that is a multiple of 8), one element otherwise.  A pass of 2-byte loads
// is 149 dependent-latency iterations for V = 151,936 - 12 us each, ten of them in a top-p call.
template <typename F>
__device__ __forceinline__ void row_read(const bf16_t* row, int V, bool vec, F f) {
  const int tid = threadIdx.x;
  if (vec) {
    const int V8 = V >> 3;
    for (int c = tid; c < V8; c += 1024) {
      const u32x4_t w = *reinterpret_cast<const u32x4_t*>(row + (size_t)c * 8);
      const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
      f(c * 8 + 0, (bf16_t)(w0 & 0xffffu)); f(c * 8 + 1, (bf16_t)(w0 >> 16));
      f(c * 8 + 2, (bf16_t)(w1 & 0xffffu)); f(c * 8 + 3, (bf16_t)(w1 >> 16));
      f(c * 8 + 4, (bf16_t)(w2 & 0xffffu)); f(c * 8 + 5, (bf16_t)(w2 >> 16));
      f(c * 8 + 6, (bf16_t)(w3 & 0xffffu)); f(c * 8 + 7, (bf16_t)(w3 >> 16));
    }
    for (int i = (V8 << 3) + tid; i < V; i += 1024) f(i, row[i]);
  } else {
    for (int i = tid; i < V; i += 1024) f(i, row[i]);
  }
}
// f(i, bits) -> the element's new bits
template <typename F>
__device__ __forceinline__ void row_update(bf16_t* row, int V, bool vec, F f) {
  const int tid = threadIdx.x;
  if (vec) {
    const int V8 = V >> 3;
    for (int c = tid; c < V8; c += 1024) {
      u32x4_t* ptr = reinterpret_cast<u32x4_t*>(row + (size_t)c * 8);
      const u32x4_t w = *ptr;
      const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
      u32x4_t o;
      o[0] = (uint32_t)f(c * 8 + 0, (bf16_t)(w0 & 0xffffu)) | ((uint32_t)f(c * 8 + 1, (bf16_t)(w0 >> 16)) << 16);
      o[1] = (uint32_t)f(c * 8 + 2, (bf16_t)(w1 & 0xffffu)) | ((uint32_t)f(c * 8 + 3, (bf16_t)(w1 >> 16)) << 16);
      o[2] = (uint32_t)f(c * 8 + 4, (bf16_t)(w2 & 0xffffu)) | ((uint32_t)f(c * 8 + 5, (bf16_t)(w2 >> 16)) << 16);
      o[3] = (uint32_t)f(c * 8 + 6, (bf16_t)(w3 & 0xffffu)) | ((uint32_t)f(c * 8 + 7, (bf16_t)(w3 >> 16)) << 16);
      if (o[0] != w0 || o[1] != w1 || o[2] != w2 || o[3] != w3) *ptr = o;
    }
    for (int i = (V8 << 3) + tid; i < V; i += 1024) row[i] = f(i, row[i]);
  } else {
    for (int i = tid; i < V; i += 1024) { const bf16_t x = row[i], y = f(i, x); if (y != x) row[i] = y; }
  }
}

// What the filters need on the device: the python scalars of the reference's closures already converted the way MLX's weak
// typing converts them (double -> float32 -> the log-probs' dtype, on the host: make_params below)
struct SamplerK {
  int use_top_p;
  float thr_top_p;      // T(1 - top_p)
  int use_min_p, min_keep;
  float log_min_p;      // T(log(min_p))
  int top_k;
  float temp;
  uint32_t seed;
  float n_sigma;        // > 0: on (fp32 statistics: the filter casts to float32 itself)
  int p_less;
  float inv_temp_t;     // T(1 / temp)
  int use_typical;
  float typical_thr;    // T(typical_p)
  float xtc_prob, xtc_thr;   // xtc_prob > 0: on; xtc_thr = T(xtc_threshold)
  const int* xtc_special;
  int n_special;
  uint32_t* sort_ws;    // typical_p: per row [2][Vp] u32 index arrays + [2][Vp] u32 payloads (sort key | log-prob bits)
  size_t sort_stride;   // u32 words per row
  int Vp;
};

constexpr int XTC_MAX_SPECIAL = 256;
constexpr int LH_WORDS = 32768 + 512;   // LDS histogram of the keys below 0x8000, one pad word per 64 keys

__global__ __launch_bounds__(1024) void sample_filter_kernel(const bf16_t* __restrict__ lp_in, int ld_in,
                                                             bf16_t* __restrict__ lp_all, int ldlp, int V,
                                                             uint32_t* __restrict__ hist_all,
                                                             const SamplerK p, const int* __restrict__ step_ptr) {
  __shared__ float red[32];
  __shared__ uint32_t s_thr_key, s_thr_keep;
  __shared__ unsigned long long s_best;
  __shared__ uint32_t scan_u[17];
  __shared__ float scan_f[17];
  __shared__ bf16_t s_special[XTC_MAX_SPECIAL];
  __shared__ uint32_t s_dig[16 * 256];
  __shared__ int s_anypos;
  extern __shared__ uint32_t lh[];               // [LH_WORDS] the lower half of the key histogram (see build_hist)           // typical-p: per-wave digit counters of the radix passes
  const int b = blockIdx.x, tid = threadIdx.x;
  bf16_t* lp = lp_all + (size_t)b * ldlp;           // scratch copy that the filters mask in place
  uint32_t* hist = hist_all + (size_t)b * ROW_W;
  const uint32_t step = (uint32_t)(step_ptr ? *step_ptr : 0);
#ifdef VLM_SAMPLE_STAMPS
  // probe build (scripts/sampler_stamps.py): phase stamps of the 100 MHz wall clock in the words of the global histogram that
  // the LDS half has made free
  int n_stamp = 0;
#define STAMP() do { __syncthreads(); if (tid == 0) hist[n_stamp] = (uint32_t)wall_clock64(); ++n_stamp; } while (0)
#else
#define STAMP() do { } while (0)
#endif
  const bf16_t* src_row = lp_in + (size_t)b * ld_in;
  const bool vec = ((uintptr_t)lp & 15) == 0;
  STAMP();   // 0
  if (vec && ((uintptr_t)src_row & 15) == 0) {
    for (int c = tid; c < (V >> 3); c += 1024) reinterpret_cast<u32x4_t*>(lp)[c] = reinterpret_cast<const u32x4_t*>(src_row)[c];
    for (int i = (V & ~7) + tid; i < V; i += 1024) lp[i] = src_row[i];
  } else {
    for (int i = tid; i < V; i += 1024) lp[i] = src_row[i];
  }
  __syncthreads();

  // The count per bf16 key.  Log-probs are <= 0: their keys are the lower 32 Ki, and that half of the histogram lives in LDS
  // (padded one word per 64 keys: a thread walks 64 CONSECUTIVE keys, lanes 65 words apart land in different banks).  Keys
  // of positive values (a caller may hand any row to a sampler) go to the global histogram - built, zeroed and read only
  // when such a value exists.  (In global memory alone: 150,000 atomics into a few hundred words of one L2, and a thread's 64
  // keys 64 cache lines apart for every lane of a load - most of a 500 us top-p call.)
  bool any_pos = false, dirty_hist = false;      // dirty_hist: the global half of the histogram holds counts (zeroed again at the end:
                                                 // between calls the workspace's histogram is all zero - the split top-p path adds into it)
  auto build_hist = [&]() {
    for (int i = tid; i < LH_WORDS; i += 1024) lh[i] = 0;
    if (tid == 0) s_anypos = 0;
    __syncthreads();
    // (-inf - every token an earlier filter removed - is counted in registers: 150,000 atomics on ONE word serialise)
    uint32_t ninf = 0;
    bool pos = false;
    row_read(lp, V, vec, [&](int, bf16_t xb) {
      const uint32_t k = bf_key(xb);
      if (k == KEY_NEG_INF) ++ninf;
      else if (k < 0x8000u) atomicAdd(&lh[k + (k >> 6)], 1u);
      else pos = true;
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ninf += __shfl_xor(ninf, o, 64);
    if ((tid & 63) == 0 && ninf) atomicAdd(&lh[KEY_NEG_INF + (KEY_NEG_INF >> 6)], ninf);
    if (pos) s_anypos = 1;
    __syncthreads();
    any_pos = s_anypos != 0;
    if (any_pos) {
      for (int i = 32768 + tid; i < 65536; i += 1024) hist[i] = 0;
      __syncthreads();
      row_read(lp, V, vec, [&](int, bf16_t xb) {
        const uint32_t k = bf_key(xb);
        if (k >= 0x8000u) atomicAdd(&hist[k], 1u);
      });
      __syncthreads();
      dirty_hist = true;
    }
  };
  auto H = [&](uint32_t k) -> uint32_t { return k < 0x8000u ? lh[k + (k >> 6)] : (any_pos ? hist[k] : 0u); };
  auto bmax = [&](float v) {
    v = wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float m = red[0];
    for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
    return m;
  };

  // ---- top-n-sigma (sample_utils.py:181-212): statistics of the float32 copy; keep x >= max - n_sigma * std (ddof 0)
  if (p.n_sigma > 0.f) {
    float m = -INFINITY, s = 0.f;
    row_read(lp, V, vec, [&](int, bf16_t xb) { const float f = bf2f(xb); m = fmaxf(m, f); s += f; });
    const float top = bmax(m);
    const float mean = block_sum(s, red) / (float)V;
    float q = 0.f;
    row_read(lp, V, vec, [&](int, bf16_t xb) { const float d = bf2f(xb) - mean; q += d * d; });
    const float sd = sqrtf(block_sum(q, red) / (float)V);
    const float thr = top - p.n_sigma * sd;          // (a row that already holds -inf: NaN statistics, nothing removed - as MLX)
    row_update(lp, V, vec, [&](int, bf16_t xb) -> bf16_t { return bf2f(xb) < thr ? NEG_INF_BF : xb; });
    __syncthreads();
  }

  // ---- p-less (sample_utils.py:215-236): keep p >= sum p^2 of softmax(x * T(1 / temp)); every op rounds to T
  if (p.p_less) {
    float m = -INFINITY;
    row_read(lp, V, vec, [&](int, bf16_t xb) { m = fmaxf(m, rbf(bf2f(xb) * p.inv_temp_t)); });
    m = bmax(m);
    float s = 0.f;
    row_read(lp, V, vec, [&](int, bf16_t xb) { s += expf(rbf(bf2f(xb) * p.inv_temp_t) - m); });
    const float den = block_sum(s, red);
    float q = 0.f;
    row_read(lp, V, vec, [&](int, bf16_t xb) {
      const float pr = rbf(expf(rbf(bf2f(xb) * p.inv_temp_t) - m) / den);
      q += rbf(pr * pr);
    });
    const float thr = rbf(block_sum(q, red));
    row_update(lp, V, vec, [&](int, bf16_t xb) -> bf16_t {
      return rbf(expf(rbf(bf2f(xb) * p.inv_temp_t) - m) / den) < thr ? NEG_INF_BF : xb;
    });
    __syncthreads();
  }

  // ---- typical-p (sample_utils.py:321-345): tokens in ascending order of |-logp - entropy| (stable: equal keys in index
  // order), kept while the cumulative probability BEFORE them is below typical_p.  The order is a real sort here (the key is
  // not monotone in logp and equal keys hold different probabilities): an LSD radix sort of the indices by the 15-bit key of
  // the non-negative bf16 value, two stable 8-bit passes.  Wave w owns a contiguous range of positions and walks it 64 at a
  // time (coalesced index loads); digit counters per wave in LDS; stability comes from the (digit, wave) order of the counter
  // scan, the iteration order inside a wave and the lane order inside an iteration (the lanes that share a digit find each
  // other with 8 ballots).  (First version: a contiguous chunk and 256 global counters per THREAD - 1.7 ms per call.)
  if (p.use_typical) {
    float acc = 0.f;
    row_read(lp, V, vec, [&](int, bf16_t xb) {
      const float lf = bf2f(xb);
      acc += rbf(rbf(expf(lf)) * lf);                 // (0 * -inf = NaN on a row that was already filtered: as MLX)
    });
    const float ent = rbf(-rbf(block_sum(acc, red)));
    // what the later passes need of a token travels WITH its index - the 15-bit sort key and the log-prob's own bits in one
    // word - so that no pass gathers from the row, and every loop has the next iteration's loads in flight before it touches
    // the counters / stores of this one (a gather + a dependent load chain per iteration was 576 us per call)
    auto payload = [&](bf16_t xb) -> uint32_t { return (((uint32_t)f2bf(fabsf(rbf(-bf2f(xb) - ent))) & 0x7fffu) << 16) | (uint32_t)xb; };
    uint32_t* idxA = p.sort_ws + (size_t)b * p.sort_stride;
    uint32_t* idxB = idxA + p.Vp;
    uint32_t* payA = idxB + p.Vp;
    uint32_t* payB = payA + p.Vp;
    const int lane = tid & 63, wave = tid >> 6;
    const int per = (((V + 15) >> 4) + 63) & ~63;                 // positions per wave: contiguous, walked 64 at a time
    const int w_lo = min(V, wave * per), w_hi = min(V, w_lo + per);
    uint32_t* my_dig = s_dig + wave * 256;
    for (int d = 0; d < 2; ++d) {
      const uint32_t *si = d == 0 ? nullptr : idxB, *sp = d == 0 ? nullptr : payB;
      uint32_t *di = d == 0 ? idxB : idxA, *dp = d == 0 ? payB : payA;
      const int sh = 16 + 8 * d;
      auto fetch = [&](int q, uint32_t& i, uint32_t& pay) {
        i = 0; pay = 0;
        if (q < w_hi) {
          if (si) { i = si[q]; pay = sp[q]; }
          else { i = (uint32_t)q; pay = payload(lp[q]); }
        }
      };
      for (int j = lane; j < 256; j += 64) my_dig[j] = 0;
      uint32_t ci, cp;
      fetch(w_lo + lane, ci, cp);
      for (int q0 = w_lo; q0 < w_hi; q0 += 64) {
        uint32_t ni, np;
        fetch(q0 + 64 + lane, ni, np);
        if (q0 + lane < w_hi) atomicAdd(&my_dig[(cp >> sh) & 255u], 1u);
        ci = ni; cp = np;
      }
      __syncthreads();
      // exclusive prefix in (digit, wave) order = the stable order: thread t owns the flattened entries 4 t .. 4 t + 3
      uint32_t v[4], mine = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int f = 4 * tid + e; v[e] = s_dig[(f & 15) * 256 + (f >> 4)]; mine += v[e]; }
      uint32_t total;
      uint32_t run = block_excl_scan<uint32_t>(mine, scan_u, &total);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int f = 4 * tid + e; s_dig[(f & 15) * 256 + (f >> 4)] = run; run += v[e]; }
      __syncthreads();
      fetch(w_lo + lane, ci, cp);
      for (int q0 = w_lo; q0 < w_hi; q0 += 64) {
        uint32_t ni, np;
        fetch(q0 + 64 + lane, ni, np);
        const bool valid = q0 + lane < w_hi;
        const uint32_t dg = valid ? (cp >> sh) & 255u : 0u;
        // the lanes of this iteration that hold the same digit (8 ballots): rank inside the group by lane = by position
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
          const unsigned long long bb = __ballot((dg >> bit) & 1u);
          same &= ((dg >> bit) & 1u) ? bb : ~bb;
        }
        if (valid) {
          const uint32_t base = my_dig[dg];
          const uint32_t at = base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
          di[at] = ci;
          dp[at] = cp;
          // (the wave's LDS operations execute in order: every lane has read before the group's last lane writes)
          if (lane == 63 - __clzll(same)) my_dig[dg] = base + (uint32_t)__popcll(same);
        }
        ci = ni; cp = np;
      }
      __syncthreads();
    }
    // the running sum of T(exp(logp)) in sorted order: wave totals, then an inclusive scan over the lanes of each iteration
    float wsum = 0.f;
    for (int q0 = w_lo; q0 < w_hi; q0 += 64) {
      const int q = q0 + lane;
      wsum += q < w_hi ? rbf(expf(bf2f((bf16_t)(payA[q] & 0xffffu)))) : 0.f;
    }
    wsum = wave_sum(wsum);
    __syncthreads();
    if (lane == 0) scan_f[wave] = wsum;
    __syncthreads();
    float run = 0.f;
    for (int w = 0; w < wave; ++w) run += scan_f[w];
    uint32_t ci = 0, cp = 0;
    if (w_lo + lane < w_hi) { ci = idxA[w_lo + lane]; cp = payA[w_lo + lane]; }
    for (int q0 = w_lo; q0 < w_hi; q0 += 64) {
      uint32_t ni = 0, np = 0;
      if (q0 + 64 + lane < w_hi) { ni = idxA[q0 + 64 + lane]; np = payA[q0 + 64 + lane]; }
      const bool valid = q0 + lane < w_hi;
      const float pe = valid ? rbf(expf(bf2f((bf16_t)(cp & 0xffffu)))) : 0.f;
      float inc = pe;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_up(inc, o, 64);
        if (lane >= o) inc += n;
      }
      const float before = rbf(rbf(run + inc) - pe);
      const float tot = __shfl(inc, 63, 64);
      // (token ci is this lane's alone: nobody else reads or writes lp[ci] now)
      if (valid && !(before < p.typical_thr)) lp[ci] = NEG_INF_BF;
      run += tot;
      ci = ni; cp = np;
    }
    __syncthreads();
  }

  // ---- top-p (sample_utils.py:289-318): ascending stable sort, keep x_i iff T(cumulative prob, inclusive) > T(1 - top_p);
  // probabilities are T(exp(x)) (typed graph: for bf16 log-probs every step rounds, tests/golden/samplers_ref.npz)
  STAMP();   // 1: row copied, earlier filters done
  if (p.use_top_p) {
    build_hist();
    STAMP();   // 2: histogram
    // each thread owns 64 consecutive keys (ascending); prefix of the probability mass over threads
    float mass = 0.f;
    for (int j = 0; j < 64; ++j) {
      const uint32_t k = tid * 64 + j, c = H(k);
      if (c) mass = fmaf((float)c, rbf(expf(bf2f(key_bf(k)))), mass);
    }
    float total;
    float cum = block_excl_scan<float>(mass, scan_f, &total);
    const float thr = p.thr_top_p;
    if (tid == 0) s_best = ~0ull;
    __syncthreads();
    STAMP();   // 3: mass per thread + scan
    // The first element (ascending key, then index) whose rounded inclusive prefix exceeds the threshold: every thread walks
    // its own keys from its exclusive prefix and proposes (key, elements of the bin left below the threshold); the smallest
    // proposal wins.  (The walk re-associates the sum the scan made; letting every thread speak - a thread that starts above
    // the threshold proposes its first key - keeps a crossing that falls between two threads.)
    {
      bool above = rbf(cum) > thr;
      for (int j = 0; j < 64; ++j) {
        const uint32_t k = tid * 64 + j, c = H(k);
        if (!c) continue;
        const float pk = rbf(expf(bf2f(key_bf(k))));
        if (pk == 0.f) continue;
        if (above) { atomicMin(&s_best, (unsigned long long)k << 32); break; }
        if (rbf(fmaf((float)c, pk, cum)) > thr) {
          // inside the bin the prefix after n elements is cum + n pk (one rounding - closer to the wide accumulator of a
          // library cumsum than n float32 additions, and monotone in n): the smallest n that crosses, by bisection.  (One
          // addition per element on a single lane was 96 us of a 250 us call: 1900 elements share the crossing key.)
          uint32_t lo_n = 1, hi_n = c;                      // invariant: n = hi_n crosses
          while (lo_n < hi_n) {
            const uint32_t mid = (lo_n + hi_n) >> 1;
            if (rbf(fmaf((float)mid, pk, cum)) > thr) hi_n = mid;
            else lo_n = mid + 1;
          }
          atomicMin(&s_best, ((unsigned long long)k << 32) | (hi_n - 1));     // hi_n - 1 elements stay below
          break;
        }
        cum = fmaf((float)c, pk, cum);
      }
    }
    __syncthreads();
    STAMP();   // 4: crossing found
    const uint32_t tk = s_best == ~0ull ? 65536u : (uint32_t)(s_best >> 32);
    const uint32_t keep = tk < 65536 ? H(tk) - (uint32_t)(s_best & 0xffffffffu) : 0;
    // (no crossing at all - T(1 - top_p) rounds to 1 for top_p < 2^-9 in bf16 and the reference then removes EVERY token;
    // a row of -inf has no sample: left unfiltered here)
    if (tk < 65536) {
      const uint32_t c = H(tk);
      // ascending stable sort: equal keys are in index order and the cumulative grows with the index,
      // so the LAST `keep` of the bin survive
      if (keep < c) mask_equal_by_rank(lp, V, vec, tk, c - keep, c, scan_u);
      STAMP(); // 5: ranks inside the crossing bin
      row_update(lp, V, vec, [&](int, bf16_t xb) -> bf16_t { return bf_key(xb) < tk ? NEG_INF_BF : xb; });
    }
    __syncthreads();
    STAMP();   // 6: masked
  }

  // ---- min-p (sample_utils.py:266-286): drop x < T(max + T(log(min_p))); min_tokens_to_keep > 1: the k largest are never
  // dropped (argpartition(kth=-k)[-k:]; ties at the k-th value: the HIGHEST indices - oracle/ops.py::apply_min_p)
  if (p.use_min_p) {
    float m = -INFINITY;
    row_read(lp, V, vec, [&](int, bf16_t xb) { m = fmaxf(m, bf2f(xb)); });
    const float thr = rbf(bmax(m) + p.log_min_p);
    bool by_rank = false;
    if (p.min_keep > 1 && p.min_keep < V) {
      build_hist();
      uint32_t cnt = 0;
      for (int j = 0; j < 64; ++j) cnt += H(65535 - (tid * 64 + j));
      uint32_t total;
      uint32_t acc = block_excl_scan<uint32_t>(cnt, scan_u, &total);
      if (tid == 0) { s_thr_key = 0; s_thr_keep = 0xffffffffu; }
      __syncthreads();
      if (acc < (uint32_t)p.min_keep && acc + cnt >= (uint32_t)p.min_keep) {
        for (int j = 0; j < 64; ++j) {
          const uint32_t k = 65535 - (tid * 64 + j), c = H(k);
          if (acc + c >= (uint32_t)p.min_keep) { s_thr_key = k; s_thr_keep = (uint32_t)p.min_keep - acc; break; }
          acc += c;
        }
      }
      __syncthreads();
      const uint32_t tk = s_thr_key, need = s_thr_keep;
      // the k-th largest value is below the threshold: the survivors are exactly the k largest
      if (need != 0xffffffffu && bf2f(key_bf(tk)) < thr) {
        by_rank = true;
        const uint32_t c = H(tk);
        if (need < c) mask_equal_by_rank(lp, V, vec, tk, c - need, c, scan_u);
        row_update(lp, V, vec, [&](int, bf16_t xb) -> bf16_t { return bf_key(xb) < tk ? NEG_INF_BF : xb; });
      }
    }
    if (!by_rank)
      row_update(lp, V, vec, [&](int, bf16_t xb) -> bf16_t { return bf2f(xb) < thr ? NEG_INF_BF : xb; });
    __syncthreads();
  }

  // ---- XTC (sample_utils.py:348-376), one row: with probability xtc_probability remove every token whose probability is
  // above the SMALLEST probability that exceeds the threshold, except the special tokens.  The draw is the counter hash at
  // an index no vocabulary entry has.
  if (p.xtc_prob > 0.f && !(hash_uniform(p.seed, step, (uint32_t)b, 0xFFFFFFFFu) > p.xtc_prob)) {
    float m = -INFINITY;
    row_read(lp, V, vec, [&](int, bf16_t xb) { m = fmaxf(m, bf2f(xb)); });
    m = bmax(m);
    float s = 0.f;
    row_read(lp, V, vec, [&](int, bf16_t xb) { s += expf(bf2f(xb) - m); });
    const float den = block_sum(s, red);
    float cand = INFINITY;
    row_read(lp, V, vec, [&](int, bf16_t xb) {
      const float pr = rbf(expf(bf2f(xb) - m) / den);
      if (pr > p.xtc_thr) cand = fminf(cand, pr);
    });
    cand = -bmax(-cand);
    if (tid < p.n_special) { const int t = p.xtc_special[tid]; s_special[tid] = (t >= 0 && t < V) ? lp[t] : (bf16_t)0; }
    __syncthreads();
    row_update(lp, V, vec, [&](int, bf16_t xb) -> bf16_t { return rbf(expf(bf2f(xb) - m) / den) > cand ? NEG_INF_BF : xb; });
    __syncthreads();
    if (tid < p.n_special) { const int t = p.xtc_special[tid]; if (t >= 0 && t < V) lp[t] = s_special[tid]; }
    __syncthreads();
  }

  // ---- top-k (sample_utils.py:169-175): keep the k largest (ties at the k-th value: lowest indices)
  const int top_k = p.top_k;
  if (top_k > 0 && top_k < V) {
    build_hist();
    // descending walk: thread t owns keys 65535 - 64 t - j
    uint32_t cnt = 0;
    for (int j = 0; j < 64; ++j) cnt += H(65535 - (tid * 64 + j));
    uint32_t total;
    uint32_t acc = block_excl_scan<uint32_t>(cnt, scan_u, &total);
    if (tid == 0) { s_thr_key = 0; s_thr_keep = 0xffffffffu; }
    __syncthreads();
    if (acc < (uint32_t)top_k && acc + cnt >= (uint32_t)top_k) {
      for (int j = 0; j < 64; ++j) {
        const uint32_t k = 65535 - (tid * 64 + j), c = H(k);
        if (acc + c >= (uint32_t)top_k) { s_thr_key = k; s_thr_keep = (uint32_t)top_k - acc; break; }
        acc += c;
      }
    }
    __syncthreads();
    const uint32_t tk = s_thr_key, keep = s_thr_keep;
    if (keep != 0xffffffffu) {
      if (keep < H(tk)) mask_equal_by_rank(lp, V, vec, tk, 0, keep, scan_u);
      row_update(lp, V, vec, [&](int, bf16_t xb) -> bf16_t { return bf_key(xb) < tk ? NEG_INF_BF : xb; });
    }
    __syncthreads();
  }

  STAMP();     // 7 (2 without top-p): filters done
#ifdef VLM_SAMPLE_STAMPS
  if (tid == 0) hist[31] = (uint32_t)n_stamp;
#else
  if (dirty_hist) {        // (uniform: set behind a barrier)
    __syncthreads();
    for (int i = 32768 + tid; i < 65536; i += 1024) hist[i] = 0;
  }
#endif
}

// ------------------------------------------------------------------------------------------
// top-p (round 5) and, since round 6, any subset of top-p / min-p / top-k (make_sampler's common chain), a row split over SPLIT_G
// workgroups.  The
// one-workgroup kernel above spends 60 of its 88 us at V = 151,936 in four passes over the 300 KB row on ONE CU (copy, histogram,
// ranks inside the crossing bin, mask) and 27 in 64-iteration loops of dependent LDS reads; here
//   A  topp_hist_kernel   (G x B workgroups)  each slice counts its keys in a private LDS histogram and adds the non-empty bins to
//                                             the row's global histogram (all zero between calls: D re-zeroes it);
//   B  topp_cross_kernel  (B workgroups)      the SAME mass / scan / crossing walk as above on the merged histogram - same thread
//                                             -> key ownership, same float operations in the same order, hence the same crossing
//                                             (key, rank) bit for bit - with a thread's 64 counts read up front;
//   C  topp_count_kernel  (G x B)             elements with the crossing key per slice;
//   D  topp_mask_kernel   (G x B)             writes the filtered row (reads the log-probs directly: no copy pass): below the
//                                             crossing key -> -inf, the crossing key's elements by their rank in index order
//                                             (prefix over the slices' counts + ballots inside the slice), re-zeroes the histogram.
// Four short launches instead of one long one, and the two passes around the filter ride in them: from logits, launch A also
// computes and writes the log-probs (topp_hist_kernel<true>: the work of logprob_argmax_kernel); launch D also makes the Gumbel
// draw over the survivors (the work of gumbel_partial_kernel).  A sampled top-p step is lse partials -> A -> B -> C -> D -> final
// argmax: six launches where the first split form had eight.  tests/test_sampler_gpu.py compares rows and tokens with the
// one-workgroup kernel (log-prob input and logits input), the oracle and the reference's golden rows.
// ------------------------------------------------------------------------------------------
constexpr int SPLIT_G = 64;
// CTL_WORDS (= 128, defined with the workspace row at the top) per row: [0] crossing key (65536: none), [1] ranks below this are masked, [2] bin count,
                                     // [3] a positive key exists, [4] / [5] smallest / largest finite key below 0x8000 (reset to
                                     // 0xffffffff / 0 by the mask kernel), [8 + s] elements with the crossing key in slice s
constexpr int PW_MAX = 7424;         // keys in the cross kernel's probability window (29 KB of LDS behind the 130 KB histogram)

__device__ __forceinline__ void slice_bounds(int V, int s, int& c_lo, int& c_hi) {     // in 8-element chunks
  const int V8 = V >> 3, cps = (V8 + SPLIT_G - 1) / SPLIT_G;
  c_lo = min(V8, s * cps);
  c_hi = min(V8, c_lo + cps);
}

// FROM_LOGITS: the launch also IS the log-prob pass of the sampled step - lp_in holds the logits, `lse_ws` the NBLK (max, sum)
// partials of lse_partial_kernel, and the slice's log-probs (bf16(logit - bf16(logsumexp)), ar.py:368: what logprob_argmax_kernel
// writes) go to `lp_out` on their way into the histogram.  A slice is exactly a block of that kernel (8 ceil(V / 512) elements),
// so the sampled step loses a launch and a second read of the row.
template <bool FROM_LOGITS>
__global__ __launch_bounds__(256) void topp_hist_kernel(const bf16_t* __restrict__ lp_in, int ld_in, int V,
                                                        uint32_t* __restrict__ hist_all, uint32_t* __restrict__ ctl_all,
                                                        const float* __restrict__ lse_ws, bf16_t* __restrict__ lp_out, int ld_out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lh[];       // [32768] counts of the keys below 0x8000 (no padding here:
                                                                      // zeroed and flushed as 16-byte pieces)
  const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const bf16_t* row = lp_in + (size_t)b * ld_in;
  uint32_t* hist = hist_all + (size_t)b * ROW_W;
  float lse = 0.f;
  if (FROM_LOGITS) {      // (the merge of logprob_argmax_kernel: lane i takes partial i, every wave redundantly)
    const int li = tid & 63;
    const float pm = lse_ws[(size_t)b * ROW_W + li * 2], ps = lse_ws[(size_t)b * ROW_W + li * 2 + 1];
    const float m = wave_max(pm);
    const float sm = wave_sum(pm > -INFINITY ? ps * expf(pm - m) : 0.f);
    lse = rbf(m + logf(sm));
  }
  const u32x4_t z = {0u, 0u, 0u, 0u};
#pragma unroll 8
  for (int i = tid; i < 8192; i += 256) reinterpret_cast<u32x4_t*>(lh)[i] = z;
  __syncthreads();
  int c_lo, c_hi;
  slice_bounds(V, s, c_lo, c_hi);
  uint32_t ninf = 0, kmin = 0xffffffffu, kmax = 0u;
  bool pos = false;
  for (int c = c_lo + tid; c < c_hi; c += 256) {
    u32x4_t w = *reinterpret_cast<const u32x4_t*>(row + (size_t)c * 8);
    if (FROM_LOGITS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t lo16 = f2bf(bf2f((bf16_t)(w[q] & 0xffffu)) - lse), hi16 = f2bf(bf2f((bf16_t)(w[q] >> 16)) - lse);
        w[q] = lo16 | (hi16 << 16);
      }
      *reinterpret_cast<u32x4_t*>(lp_out + (size_t)b * ld_out + (size_t)c * 8) = w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t k = bf_key((bf16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu)));
      if (k == KEY_NEG_INF) ++ninf;
      else if (k < 0x8000u) { atomicAdd(&lh[k], 1u); kmin = min(kmin, k); kmax = max(kmax, k); }
      else { atomicAdd(&hist[k], 1u); pos = true; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ninf += __shfl_xor(ninf, o, 64);
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o, 64));
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o, 64));
  }
  if ((tid & 63) == 0) {
    if (ninf) atomicAdd(&lh[KEY_NEG_INF], ninf);
    if (kmin <= kmax) {      // the populated range of the finite non-positive keys (the cross kernel's probability window)
      atomicMin(&ctl_all[(size_t)b * ROW_W + 4], kmin);
      atomicMax(&ctl_all[(size_t)b * ROW_W + 5], kmax);
    }
  }
  if (pos) ctl_all[(size_t)b * ROW_W + 3] = 1u;
  __syncthreads();
  // flush: 4 bins per 16-byte read, 4 reads in flight; a slice of 2400 elements touches a few hundred bins
#pragma unroll 1
  for (int i0 = tid; i0 < 8192; i0 += 1024) {
    u32x4_t q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = reinterpret_cast<const u32x4_t*>(lh)[i0 + 256 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (q[u][e]) atomicAdd(&hist[4 * (i0 + 256 * u) + e], q[u][e]);
  }
}

// Round 6: the launch also folds min-p (min_tokens_to_keep = 1) and top-k into the rule it leaves for the mask launch.  The three
// filters run in the reference's order top-p -> min-p -> top-k (sample_utils.py:69-76), each on the row the previous one left, and each
// is monotone in (value, then index): after any prefix of them the survivors are "every element above a key, plus a window of ranks
// (index order) inside that key's bin".  So the chain is evaluated ON THE HISTOGRAM - top-p's crossing as before; min-p = the smallest
// key whose value reaches T(max + T(log min_p)) (sample_utils.py:266-286); top-k = a descending count over the surviving bins
// (sample_utils.py:169-175: ties at the k-th value keep the LOWEST indices of what is left of that bin) - and ctl carries the final
// (key, [lo, hi)) for the count / mask launches.  Same survivors, bit for bit, as sample_filter_kernel's three passes over the row.
__global__ __launch_bounds__(1024) void topp_cross_kernel(const uint32_t* __restrict__ hist_all, uint32_t* __restrict__ ctl_all,
                                                          float thr, int use_top_p, int use_min_p, float log_min_p, int top_k, int V) {
  __shared__ float scan_f[17];
  __shared__ uint32_t scan_u2[17];
  __shared__ uint32_t s_kmax, s_k2, s_keep2;
  __shared__ unsigned long long s_best;
  extern __shared__ uint32_t lh[];               // [LH_WORDS] the lower half of the merged histogram, padded as above
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint32_t* hist = hist_all + (size_t)b * ROW_W;
  uint32_t* ctl = ctl_all + (size_t)b * ROW_W;
#ifdef VLM_TOPP_STAMPS
  int n_st = 0;
#define XST() do { __syncthreads(); if (tid == 0) ctl[80 + n_st] = (uint32_t)wall_clock64(); ++n_st; } while (0)
#else
#define XST() do { } while (0)
#endif
  XST();
#pragma unroll 1
  for (int k0 = tid; k0 < 32768; k0 += 8192) {        // 8 independent loads per trip (a dependent loop of 32 L2 round trips
    uint32_t v[8];                                     // was most of this launch)
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = hist[k0 + 1024 * u];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int k = k0 + 1024 * u; lh[k + (k >> 6)] = v[u]; }
  }
  const bool any_pos = ctl[3] != 0;
  // p(k) = T(exp(value of key k)) is needed for every non-empty bin, twice, by the thread that owns the bin - 64 consecutive keys
  // per thread, so the few hundred populated bins of a row of log-probs sit in a handful of threads and their exps ran one after
  // the other (most of this launch).  The populated range [kmin, kmax] (topp_hist_kernel) almost always fits the LDS left beside
  // the histogram (PW_MAX keys = 58 binades): every thread computes the p of a few keys of that window first.  Same expf, same
  // rounding - the walk below reads the value instead of computing it.  A row with positive keys, or a wider range, computes on
  // the fly as the one-workgroup kernel does.
  float* pw = reinterpret_cast<float*>(lh + LH_WORDS);
  const uint32_t kmin = ctl[4] & ~63u, kmax = ctl[5] | 63u;      // whole 64-key blocks: a thread's keys are all inside or all outside
  const bool windowed = !any_pos && ctl[4] <= ctl[5] && kmax - kmin < (uint32_t)PW_MAX;
  if (tid == 0) { s_best = ~0ull; s_kmax = 0u; s_k2 = 65536u; s_keep2 = 0u; }
  __syncthreads();
  XST();
  if (windowed && use_top_p) {
    for (uint32_t k = kmin + tid; k <= kmax; k += 1024) pw[k - kmin] = lh[k + (k >> 6)] ? rbf(expf(bf2f(key_bf(k)))) : 0.f;
    __syncthreads();
  }
  XST();
  auto prob = [&](uint32_t k) -> float {
    return (windowed && k >= kmin && k <= kmax) ? pw[k - kmin] : rbf(expf(bf2f(key_bf(k))));
  };
  // This thread's 64 consecutive keys (ascending): mass = sum of count x p in key order, then (after the scan over the threads)
  // the walk to the first key whose rounded inclusive prefix exceeds the threshold.  The one-workgroup kernel does both with a
  // dependent LDS read and a branch per key (27 of its 88 us).  In a windowed row a thread's block of keys lies wholly inside
  // the window - counts and probabilities are read 16 at a time and every key takes the same few instructions: an empty bin has
  // p = 0 in the window and fmaf(c, 0, x) = x exactly, so the sums are the ones the skipping loops make - or wholly outside,
  // where only the -inf bin can be populated (p = 0): nothing to add, nothing to propose.
  auto H = [&](uint32_t k) -> uint32_t { return k < 0x8000u ? lh[k + (k >> 6)] : (any_pos ? hist[k] : 0u); };
  const uint32_t k0 = (uint32_t)tid * 64u;
  const bool mine = windowed && k0 >= kmin && k0 <= kmax;
  float mass = 0.f;
  XST();
  if (!use_top_p) {
    // (no nucleus filter in the chain: nothing to sum)
  } else if (windowed) {
    if (mine) {
#pragma unroll 1
      for (int j0 = 0; j0 < 64; j0 += 16) {
        uint32_t cc[16];
        float pp[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { cc[j] = lh[k0 + (k0 >> 6) + j0 + j]; pp[j] = pw[k0 - kmin + j0 + j]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) mass = fmaf((float)cc[j], pp[j], mass);
      }
    }
  } else {
    for (int j = 0; j < 64; ++j) {
      const uint32_t k = k0 + j, c = H(k);
      if (c) mass = fmaf((float)c, prob(k), mass);
    }
  }
  XST();
  float total;
  float cum = block_excl_scan<float>(mass, scan_f, &total);
  XST();
  if (use_top_p) {
    const bool above = rbf(cum) > thr;
    auto cross_in_bin = [&](uint32_t k, uint32_t c, float pk, float base) {     // the smallest n with T(base + n pk) > thr, by bisection
      uint32_t lo_n = 1, hi_n = c;                        // invariant: n = hi_n crosses
      while (lo_n < hi_n) {
        const uint32_t mid = (lo_n + hi_n) >> 1;
        if (rbf(fmaf((float)mid, pk, base)) > thr) hi_n = mid;
        else lo_n = mid + 1;
      }
      atomicMin(&s_best, ((unsigned long long)k << 32) | (hi_n - 1));     // hi_n - 1 elements stay below
    };
    if (windowed) {
      if (mine) {
        // branch-free over the block: the first key with c != 0 and p != 0 (the proposal of a thread that starts above the
        // threshold), and the first key whose inclusive prefix crosses, with the prefix it started from
        uint32_t first_any = 64u, first_x = 64u, c_x = 0u;
        float p_x = 0.f, base_x = 0.f;
#pragma unroll 1
        for (int j0 = 0; j0 < 64; j0 += 16) {
          uint32_t cc[16];
          float pp[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) { cc[j] = lh[k0 + (k0 >> 6) + j0 + j]; pp[j] = pw[k0 - kmin + j0 + j]; }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const bool live = cc[j] != 0u && pp[j] != 0.f;
            const float nxt = fmaf((float)cc[j], pp[j], cum);
            const bool x = live && first_x == 64u && rbf(nxt) > thr;
            first_any = (live && first_any == 64u) ? (uint32_t)(j0 + j) : first_any;
            if (x) { first_x = (uint32_t)(j0 + j); c_x = cc[j]; p_x = pp[j]; base_x = cum; }
            cum = (first_x == 64u) ? nxt : cum;            // (the walk stops at its crossing)
          }
        }
        if (above) {
          if (first_any != 64u) atomicMin(&s_best, (unsigned long long)(k0 + first_any) << 32);
        } else if (first_x != 64u) {
          cross_in_bin(k0 + first_x, c_x, p_x, base_x);
        }
      }
    } else {
      for (int j = 0; j < 64; ++j) {
        const uint32_t k = k0 + j, c = H(k);
        if (!c) continue;
        const float pk = prob(k);
        if (pk == 0.f) continue;
        if (above) { atomicMin(&s_best, (unsigned long long)k << 32); break; }
        if (rbf(fmaf((float)c, pk, cum)) > thr) { cross_in_bin(k, c, pk, cum); break; }
        cum = fmaf((float)c, pk, cum);
      }
    }
  }
  __syncthreads();
  XST();
  // The rule after top-p, in registers (every thread the same): key, window [lo, hi) of ranks inside its bin.  No nucleus filter /
  // no crossing (the reference then removes EVERY token; a row of -inf has no sample: left unfiltered): key 0 with its whole bin =
  // nothing is removed.
  const unsigned long long best = s_best;
  const bool crossed = use_top_p && best != ~0ull;
  uint32_t r_key = crossed ? (uint32_t)(best >> 32) : 0u;
  uint32_t r_lo = crossed ? (uint32_t)(best & 0xffffffffu) : 0u;       // ascending stable sort: the FIRST `lo` of the bin go
  uint32_t r_hi = H(r_key);
  // every finite value of the row has a key below 0x8000 and topp_hist_kernel left their exact range: the chain needs no search
  const bool finite_neg = !any_pos && ctl[4] <= ctl[5];
  const uint32_t f_lo = ctl[4], f_hi = ctl[5];
  // what is left of bin k under the current rule
  auto A = [&](uint32_t k) -> uint32_t { return k < r_key ? 0u : (k == r_key ? r_hi - r_lo : H(k)); };
  if (use_min_p) {
    // the row's maximum = the largest bin with a survivor (top-p never empties the top bin); thr = T(max + T(log(min_p))) is a
    // T value: the smallest key whose value is not below it is the key of thr itself (-0 / +0: neither is below 0)
    uint32_t top = f_hi;
    if (!finite_neg) {
      for (int j = 63; j >= 0; --j)
        if (A(k0 + j)) { atomicMax(&s_kmax, k0 + j); break; }
      __syncthreads();
      top = s_kmax;
    }
    const float thr_m = rbf(bf2f(key_bf(top)) + log_min_p);
    if (thr_m == thr_m) {
      const uint32_t km = thr_m == 0.f ? 0x7fffu : bf_key((bf16_t)(__float_as_uint(thr_m) >> 16));
      if (km > r_key) { r_key = km; r_lo = 0u; r_hi = H(km); }
    }
  }
  if (top_k > 0 && top_k < V) {
    // descending count over what is left: thread t owns the 64 keys from kb + 63 down to kb (as the one-workgroup kernel).  In a
    // finite_neg row only the blocks between the rule's key and the largest populated key can hold a survivor (the -inf bin can
    // only be the k-th value when fewer than k tokens are left, where the filter removes nothing).
    const uint32_t kb = (1023u - (uint32_t)tid) * 64u;
    const bool live = !finite_neg || (kb + 63u >= max(r_key, f_lo) && kb <= f_hi);
    uint32_t cnt = 0;
    if (live) {
#pragma unroll 16
      for (int j = 0; j < 64; ++j) cnt += A(kb + 63u - j);
    }
    uint32_t tot_u;
    uint32_t acc = block_excl_scan<uint32_t>(cnt, scan_u2, &tot_u);
    if (live && acc < (uint32_t)top_k && acc + cnt >= (uint32_t)top_k) {
      uint32_t found = 64u, keep = 0u;
#pragma unroll 16
      for (int j = 0; j < 64; ++j) {
        const uint32_t c = A(kb + 63u - j);
        const bool x = found == 64u && acc + c >= (uint32_t)top_k;
        if (x) { found = (uint32_t)j; keep = (uint32_t)top_k - acc; }
        acc += c;
      }
      s_k2 = kb + 63u - found; s_keep2 = keep;
    }
    __syncthreads();
    const uint32_t k2 = s_k2, keep2 = s_keep2;
    if (k2 < 65536u) {
      // ties at the k-th value: the lowest indices of what is left of that bin
      if (k2 == r_key) r_hi = r_lo + keep2;
      else { r_key = k2; r_lo = 0u; r_hi = keep2; }
    }
  }
  if (tid == 0) {
    const uint32_t c = H(r_key);
    ctl[0] = r_key; ctl[1] = r_lo; ctl[2] = c; ctl[6] = r_hi;
    ctl[7] = (r_lo > 0u || r_hi < c) ? 1u : 0u;            // the bin is cut: the count / mask launches rank its elements
  }
}

__device__ __forceinline__ uint32_t hits8(const u32x4_t& w, uint32_t tb) {
  const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
  return (uint32_t)((w0 & 0xffffu) == tb) | (uint32_t)((w0 >> 16) == tb) << 1 | (uint32_t)((w1 & 0xffffu) == tb) << 2 |
         (uint32_t)((w1 >> 16) == tb) << 3 | (uint32_t)((w2 & 0xffffu) == tb) << 4 | (uint32_t)((w2 >> 16) == tb) << 5 |
         (uint32_t)((w3 & 0xffffu) == tb) << 6 | (uint32_t)((w3 >> 16) == tb) << 7;
}

__global__ __launch_bounds__(256) void topp_count_kernel(const bf16_t* __restrict__ lp_in, int ld_in, int V,
                                                         uint32_t* __restrict__ ctl_all) {
  __shared__ uint32_t red[4];
  const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  uint32_t* ctl = ctl_all + (size_t)b * ROW_W;
  const uint32_t tk = ctl[0];
  uint32_t cnt = 0;
  if (ctl[7]) {
    const bf16_t* row = lp_in + (size_t)b * ld_in;
    const uint32_t tb = key_bf(tk);
    int c_lo, c_hi;
    slice_bounds(V, s, c_lo, c_hi);
    for (int c = c_lo + tid; c < c_hi; c += 256) cnt += (uint32_t)__popc(hits8(*reinterpret_cast<const u32x4_t*>(row + (size_t)c * 8), tb));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) ctl[8 + s] = red[0] + red[1] + red[2] + red[3];
}

// ... and the draw: z_i = x_i / temp - log(-log(u_i)) over the survivors as they are written (gumbel_partial_kernel's arithmetic,
// its index-keyed RNG and its lowest-index tie rule; the best (z, i) of the slice lands where that kernel's block would leave
// it), so the filtered row is not read back by a launch of its own.
__global__ __launch_bounds__(256) void topp_mask_kernel(const bf16_t* __restrict__ lp_in, int ld_in, bf16_t* __restrict__ out_all,
                                                        int ldo, int V, uint32_t* __restrict__ hist_all, uint32_t* __restrict__ ctl_all,
                                                        float temp, uint32_t seed, const int* __restrict__ step_ptr,
                                                        float* __restrict__ cand_v, int* __restrict__ cand_i) {
  __shared__ uint32_t wcnt[4];
  __shared__ float sv[4];
  __shared__ int si[4];
  const uint32_t step = (uint32_t)(step_ptr ? *step_ptr : 0);
  const float it = 1.0f / temp;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  const int s = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t* ctl = ctl_all + (size_t)b * ROW_W;
  const bf16_t* row = lp_in + (size_t)b * ld_in;
  bf16_t* out = out_all + (size_t)b * ldo;
  const uint32_t tk = ctl[0], drop = ctl[1], keep_hi = ctl[6];      // elements of bin tk survive with rank in [drop, keep_hi)
  const bool ranked = ctl[7] != 0u;
  const uint32_t tb = key_bf(min(tk, 65535u));
  int c_lo, c_hi;
  slice_bounds(V, s, c_lo, c_hi);
  // wave w owns a contiguous quarter of the slice's chunks (index order: slices, then waves, then iterations, then lanes)
  const int per = ((c_hi - c_lo + 3) >> 2), w_lo = min(c_hi, c_lo + wave * per), w_hi = min(c_hi, w_lo + per);
  uint32_t rank = 0;
  if (ranked) {
    for (int sp = 0; sp < s; ++sp) rank += ctl[8 + sp];
    uint32_t cnt = 0;
    for (int c = w_lo + lane; c < w_hi; c += 64) cnt += (uint32_t)__popc(hits8(*reinterpret_cast<const u32x4_t*>(row + (size_t)c * 8), tb));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) wcnt[wave] = cnt;
    __syncthreads();
    for (int w = 0; w < wave; ++w) rank += wcnt[w];
  }
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int c0 = w_lo; c0 < w_hi; c0 += 64) {
    const int c = c0 + lane;
    const bool in = c < w_hi;
    u32x4_t w = {0u, 0u, 0u, 0u};
    if (in) w = *reinterpret_cast<const u32x4_t*>(row + (size_t)c * 8);
    uint32_t wv[4] = {w[0], w[1], w[2], w[3]};
    uint32_t m = 0;
    if (ranked) {
      m = in ? hits8(w, tb) : 0u;
      const uint32_t n = (uint32_t)__popc(m);
      uint32_t before = 0, tot = 0;
#pragma unroll
      for (int bit = 0; bit < 4; ++bit) {
        const unsigned long long bb = __ballot((n >> bit) & 1u);
        before += (uint32_t)__popcll(bb & lt) << bit;
        tot += (uint32_t)__popcll(bb) << bit;
      }
      uint32_t r = rank + before;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if ((m >> e) & 1u) {
          if (r < drop || r >= keep_hi) wv[e >> 1] = (e & 1) ? (wv[e >> 1] & 0x0000ffffu) | ((uint32_t)NEG_INF_BF << 16) : (wv[e >> 1] & 0xffff0000u) | NEG_INF_BF;
          ++r;
        }
      rank += tot;
    }
    if (in) {
      if (tk < 65536u) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t bits = (e & 1) ? (wv[e >> 1] >> 16) : (wv[e >> 1] & 0xffffu);
          if (bf_key((bf16_t)bits) < tk)
            wv[e >> 1] = (e & 1) ? (wv[e >> 1] & 0x0000ffffu) | ((uint32_t)NEG_INF_BF << 16) : (wv[e >> 1] & 0xffff0000u) | NEG_INF_BF;
        }
      }
      u32x4_t o;
      o[0] = wv[0]; o[1] = wv[1]; o[2] = wv[2]; o[3] = wv[3];
      *reinterpret_cast<u32x4_t*>(out + (size_t)c * 8) = o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bf16_t xb = (bf16_t)((e & 1) ? (wv[e >> 1] >> 16) : (wv[e >> 1] & 0xffffu));
        if (xb == NEG_INF_BF) continue;                     // (a removed token cannot win: -inf + gumbel = -inf)
        const int i = c * 8 + e;
        const float zg = __builtin_fmaf(bf2f(xb), it, -logf(-logf(hash_uniform(seed, step, (uint32_t)b, (uint32_t)i))));      // (one fma,
                                                                                  // as hipcc contracts gumbel_partial_kernel's x * it + g)
        if (zg > best || (zg == best && i < besti)) { best = zg; besti = i; }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if (lane == 0) { sv[wave] = best; si[wave] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < besti)) { best = sv[w]; besti = si[w]; }
    cand_v[(size_t)b * ROW_W + s] = best;
    cand_i[(size_t)b * ROW_W + s] = besti;
  }
  // the row's histogram goes back to all-zero for the next call (1024 words per slice), the positive-key flag with it
  uint32_t* hist = hist_all + (size_t)b * ROW_W + (size_t)s * 1024;
  const u32x4_t z = {0u, 0u, 0u, 0u};
  reinterpret_cast<u32x4_t*>(hist)[tid] = z;
  if (s == 0 && tid == 0) { ctl[3] = 0u; ctl[4] = 0xffffffffu; ctl[5] = 0u; }
}

// categorical(logprobs / temp) (sample_utils.py:385-387) by Gumbel-max with the counter hash RNG over the (filtered) row:
// z_i = x_i / temp - log(-log(u_i)), u_i = hash(seed, step, row, i); the token is argmax z (lowest index on a tie).  NBLK
// workgroups per row leave their best (z, i) where the greedy path leaves its candidates and argmax_final_kernel picks the
// winner.  (Inside the one-workgroup-per-row filter kernel this pass was 45-52 us of hash + two logs per element on ONE CU.)
__global__ __launch_bounds__(256) void gumbel_partial_kernel(const bf16_t* __restrict__ lp_all, int ldlp, int V, float temp,
                                                             uint32_t seed, const int* __restrict__ step_ptr,
                                                             float* __restrict__ cand_v, int* __restrict__ cand_i) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int b = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const uint32_t step = (uint32_t)(step_ptr ? *step_ptr : 0);
  const int per = ((V + NBLK - 1) / NBLK + 7) & ~7;
  const int lo = blk * per, hi = min(V, lo + per);
  const bf16_t* row = lp_all + (size_t)b * ldlp;
  const float it = 1.0f / temp;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int i = lo + tid; i < hi; i += 256) {
    const bf16_t xb = row[i];
    if (xb == NEG_INF_BF) continue;                     // (a removed token cannot win: -inf + gumbel = -inf)
    const float u = hash_uniform(seed, step, (uint32_t)b, (uint32_t)i);
    const float z = __builtin_fmaf(bf2f(xb), it, -logf(-logf(u)));      // x / temp + g as ONE fma (what hipcc's contraction made of the
                                                                        // two statements before; written out so that topp_mask_kernel's copy
                                                                        // of the draw cannot drift from it)
    if (z > best || (z == best && i < besti)) { best = z; besti = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < besti)) { best = sv[w]; besti = si[w]; }
    cand_v[(size_t)b * ROW_W + blk] = best;
    cand_i[(size_t)b * ROW_W + blk] = besti;
  }
}

// float32 -> bf16 value, round to nearest even (host side of the weak-typed scalars)
inline float host_rbf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return f;
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&f, &u, 4);
  return f;
}

}  // namespace

extern "C" size_t vlm_sample_workspace_bytes(int B) {
  return (size_t)B * ROW_W * sizeof(uint32_t) + 256;
}

// typical_p's sort: per row two (index, payload) array pairs of Vp = V rounded up to 1024 entries (the digit counters live in LDS)
extern "C" size_t vlm_sample_sort_workspace_bytes(int B, int V) {
  if (B <= 0 || V <= 0) return 0;
  const size_t Vp = ((size_t)V + 1023) & ~(size_t)1023;
  return (size_t)B * 4 * Vp * sizeof(uint32_t);
}

// workspace layout: 256 B = arrival ticket of the fused greedy tail (must be zero at allocation; the kernel re-arms it; at a
// fixed offset so that a step over the first B' < B rows of a state finds the same word) + at byte 4 the count of rows whose
// argmax found no candidate (an all-NaN logits row; the token is then 0) |
// then one ROW_W-word block per row (see ROW_W at the top): lse partials | cand_v | cand_i | hist (all zero between calls) |
// control words of the split top-p path (zero at allocation) - every array at a fixed offset inside the row's block
static thread_local int t_last_launches = 0;   // kernels enqueued by the last sample_ex_impl of this thread (engine statistics)
VLM_INTERNAL int vlm_sample_last_launches(void) { return t_last_launches; }

static int sample_ex_impl(const void* logits, int ld, int B, int V, void* logprobs, void* scratch, int ldlp, void* tok,
                          void* workspace, const vlm_sampler_params* sp, const void* step_ptr, void* stream, const SampleTail* tail) {
  if (!logits || !tok || !workspace || !sp || B <= 0 || V <= 0) return VLM_ERR_ARG;
  const double temperature = sp->temperature;
  if (!(temperature >= 0.0)) return VLM_ERR_ARG;
  const bool lp_given = temperature > 0.0 && sp->input_is_logprobs;     // the sampler closure's own contract: log-probs in
  if (temperature > 0.0 && (!scratch || (!logprobs && !lp_given))) return VLM_ERR_ARG;
  SamplerK k{};
  if (temperature > 0.0) {
    // the checks of the reference's closures (sample_utils.py:160-165, 200-203, 253-260, 323-326, 363-370)
    if (sp->min_p < 0.0 || sp->min_p > 1.0 || sp->min_tokens_to_keep < 1 || sp->top_n_sigma < 0.0 || sp->top_k < 0) return VLM_ERR_ARG;
    if (sp->xtc_probability < 0.0 || sp->xtc_probability > 1.0) return VLM_ERR_ARG;
    k.use_top_p = sp->top_p > 0.0 && sp->top_p < 1.0;
    k.thr_top_p = host_rbf((float)(1.0 - sp->top_p));
    k.use_min_p = sp->min_p != 0.0;
    k.min_keep = sp->min_tokens_to_keep;
    k.log_min_p = k.use_min_p ? host_rbf((float)log(sp->min_p)) : 0.f;
    k.top_k = sp->top_k;
    k.temp = (float)temperature;
    k.seed = sp->seed;
    k.n_sigma = (float)sp->top_n_sigma;
    k.p_less = sp->p_less != 0;
    k.inv_temp_t = host_rbf((float)(1.0 / temperature));
    k.use_typical = sp->typical_p > 0.0 && sp->typical_p < 1.0;
    k.typical_thr = host_rbf((float)sp->typical_p);
    k.xtc_prob = (float)sp->xtc_probability;
    if (k.xtc_prob > 0.f) {
      if (sp->xtc_threshold < 0.0 || sp->xtc_threshold > 0.5) return VLM_ERR_ARG;
      // apply_xtc's minimum runs over the WHOLE array and its draw is one scalar per call (sample_utils.py:371-376): built for
      // the one-row call of generate_step
      if (B != 1) return VLM_ERR_SHAPE;
      if (sp->n_xtc_special < 0 || sp->n_xtc_special > XTC_MAX_SPECIAL || (sp->n_xtc_special > 0 && !sp->xtc_special_tokens))
        return VLM_ERR_ARG;
      k.xtc_thr = host_rbf((float)sp->xtc_threshold);
      k.xtc_special = (const int*)sp->xtc_special_tokens;
      k.n_special = sp->n_xtc_special;
    }
    if (k.use_typical) {
      if (!sp->sort_workspace) return VLM_ERR_ARG;
      k.Vp = (V + 1023) & ~1023;
      k.sort_stride = 4 * (size_t)k.Vp;
      k.sort_ws = (uint32_t*)sp->sort_workspace;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  t_last_launches = 0;
  float* ws = (float*)((char*)workspace + 256);
  float* cand_v = ws + ROW_CV;
  int* cand_i = (int*)(ws + ROW_CI);
  uint32_t* hist = (uint32_t*)(ws + ROW_HIST);
  // top-p / min-p / top-k over a 16-byte-aligned row of a real vocabulary: the row split over SPLIT_G workgroups
  static const bool split_env = [] { const char* e = getenv("VLM_SAMPLE_SPLIT"); return !e || atoi(e) != 0; }();   // A/B knob
  // (round 6: any subset of top-p / min-p with min_tokens_to_keep = 1 / top-k - the chain is folded into the crossing launch)
  const bool split = temperature != 0.0 && split_env && (lp_given || logprobs) && (k.use_top_p || k.use_min_p || (k.top_k > 0 && k.top_k < V)) &&
                     (!k.use_min_p || k.min_keep == 1) && !(k.n_sigma > 0.f) && !k.p_less &&
                     !k.use_typical && !(k.xtc_prob > 0.f) && V % 8 == 0 && V >= 8192 && (lp_given ? ld : ldlp) % 8 == 0 && ldlp % 8 == 0 &&
                     ((uintptr_t)(lp_given ? logits : logprobs) & 15) == 0 && ((uintptr_t)scratch & 15) == 0;
  // ... and from logits: the log-prob pass runs inside the histogram launch (topp_hist_kernel<true>)
  const bool fused_lp = split && !lp_given && ld % 8 == 0 && ((uintptr_t)logits & 15) == 0;
  if (!lp_given) {
    hipLaunchKernelGGL(lse_partial_kernel, dim3(NBLK, B), dim3(256), 0, st, (const bf16_t*)logits, ld, V, ws);
    VLM_CHECK_LAUNCH(); ++t_last_launches;
    if (!fused_lp) {
      hipLaunchKernelGGL(logprob_argmax_kernel, dim3(NBLK, B), dim3(256), 0, st, (const bf16_t*)logits, ld, V, ws,
                         (bf16_t*)logprobs, ldlp, cand_v, cand_i);
      VLM_CHECK_LAUNCH(); ++t_last_launches;
    }
  }
  if (temperature == 0.0) {
    hipLaunchKernelGGL(argmax_final_kernel, dim3(B), dim3(64), 0, st, cand_v, cand_i, (int*)tok);
  } else {
    constexpr int LDS = LH_WORDS * (int)sizeof(uint32_t);       // 130 KB of the CU's 160 (one workgroup per CU)
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&sample_filter_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (attr != hipSuccess) return VLM_ERR_HIP + (int)attr;
    const bf16_t* row_in = lp_given ? (const bf16_t*)logits : (const bf16_t*)logprobs;
    int ld_in = lp_given ? ld : ldlp;
    const bool any_filter = k.use_top_p || k.use_min_p || k.top_k > 0 || k.n_sigma > 0.f || k.p_less || k.use_typical || k.xtc_prob > 0.f;
    if (split) {
      static const hipError_t attr_a = hipFuncSetAttribute(reinterpret_cast<const void*>(&topp_hist_kernel<false>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      static const hipError_t attr_a2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&topp_hist_kernel<true>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (attr_a2 != hipSuccess) return VLM_ERR_HIP + (int)attr_a2;
      constexpr int LDS_B = LDS + PW_MAX * (int)sizeof(float);       // histogram + probability window: 159.7 KB of the CU's 160
      static const hipError_t attr_b = hipFuncSetAttribute(reinterpret_cast<const void*>(&topp_cross_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
      if (attr_a != hipSuccess || attr_b != hipSuccess) return VLM_ERR_HIP + (int)(attr_a != hipSuccess ? attr_a : attr_b);
      uint32_t* ctl = (uint32_t*)(ws + ROW_CTL);
      if (fused_lp)
        hipLaunchKernelGGL(topp_hist_kernel<true>, dim3(SPLIT_G, B), dim3(256), LDS, st, (const bf16_t*)logits, ld, V, hist, ctl,
                           (const float*)ws, (bf16_t*)logprobs, ldlp);
      else
        hipLaunchKernelGGL(topp_hist_kernel<false>, dim3(SPLIT_G, B), dim3(256), LDS, st, row_in, ld_in, V, hist, ctl,
                           (const float*)nullptr, (bf16_t*)nullptr, 0);
      VLM_CHECK_LAUNCH(); ++t_last_launches;
      hipLaunchKernelGGL(topp_cross_kernel, dim3(B), dim3(1024), LDS_B, st, (const uint32_t*)hist, ctl, k.thr_top_p, k.use_top_p, k.use_min_p,
                         k.log_min_p, k.top_k, V);
      VLM_CHECK_LAUNCH(); ++t_last_launches;
      hipLaunchKernelGGL(topp_count_kernel, dim3(SPLIT_G, B), dim3(256), 0, st, row_in, ld_in, V, ctl);
      VLM_CHECK_LAUNCH(); ++t_last_launches;
      hipLaunchKernelGGL(topp_mask_kernel, dim3(SPLIT_G, B), dim3(256), 0, st, row_in, ld_in, (bf16_t*)scratch, ldlp, V, hist, ctl,
                         k.temp, k.seed, (const int*)step_ptr, cand_v, cand_i);
      VLM_CHECK_LAUNCH(); ++t_last_launches;
      if (tail) hipLaunchKernelGGL(argmax_final_advance_kernel, dim3(1), dim3(256), 0, st, cand_v, cand_i, (int*)tok, B, V, *tail);
      else hipLaunchKernelGGL(argmax_final_kernel, dim3(B), dim3(64), 0, st, cand_v, cand_i, (int*)tok);
      VLM_CHECK_LAUNCH(); ++t_last_launches;
      return VLM_OK;
    } else if (any_filter) {
      hipLaunchKernelGGL(sample_filter_kernel, dim3(B), dim3(1024), LDS, st, row_in, ld_in, (bf16_t*)scratch, ldlp, V, hist, k,
                         (const int*)step_ptr);
      VLM_CHECK_LAUNCH(); ++t_last_launches;
      row_in = (const bf16_t*)scratch;
      ld_in = ldlp;
    }
    // (no filter: the draw runs over the log-probs themselves; `scratch` is left untouched)
    hipLaunchKernelGGL(gumbel_partial_kernel, dim3(NBLK, B), dim3(256), 0, st, row_in, ld_in, V, k.temp, k.seed,
                       (const int*)step_ptr, cand_v, cand_i);
    VLM_CHECK_LAUNCH(); ++t_last_launches;
    if (tail) hipLaunchKernelGGL(argmax_final_advance_kernel, dim3(1), dim3(256), 0, st, cand_v, cand_i, (int*)tok, B, V, *tail);
    else hipLaunchKernelGGL(argmax_final_kernel, dim3(B), dim3(64), 0, st, cand_v, cand_i, (int*)tok);
  }
  VLM_CHECK_LAUNCH(); ++t_last_launches;
  return VLM_OK;
}

extern "C" int vlm_sample_ex(const void* logits, int ld, int B, int V, void* logprobs, void* scratch, int ldlp, void* tok,
                             void* workspace, const vlm_sampler_params* sp, const void* step_ptr, void* stream) {
  return sample_ex_impl(logits, ld, B, V, logprobs, scratch, ldlp, tok, workspace, sp, step_ptr, stream, nullptr);
}

// vlm_sample (temperature > 0) + vlm_decode_advance + the next step's vlm_embed_gather, for the engine's captured step (internal.h)
VLM_INTERNAL int vlm_sample_advance(const void* logits, int ld, int B, int V, void* logprobs, void* scratch, int ldlp, void* tok,
                                    void* workspace, float temperature, float top_p, float min_p, int top_k, unsigned seed,
                                    void* ctx, void* pos, void* out_ring, int ring_len, void* step, const void* embed, void* h,
                                    int D, int ldh, void* stream) {
  if (!(temperature > 0.f) || !ctx || !pos || !step || !embed || !h) return VLM_ERR_ARG;
  if (B > 64 || D % 8 || ldh % 8 || (out_ring && ring_len <= 0)) return VLM_ERR_SHAPE;
  vlm_sampler_params sp{};
  sp.temperature = temperature;
  sp.top_p = top_p;
  sp.min_p = min_p;
  sp.min_tokens_to_keep = 1;
  sp.top_k = top_k;
  sp.typical_p = 1.0;
  sp.seed = seed;
  const SampleTail t{(int*)ctx, (int*)pos, (int*)out_ring, (int*)step, ring_len, (const bf16_t*)embed, (bf16_t*)h, D, ldh};
  return sample_ex_impl(logits, ld, B, V, logprobs, scratch, ldlp, tok, workspace, &sp, step, stream, &t);
}

extern "C" int vlm_sample(const void* logits, int ld, int B, int V, void* logprobs, void* scratch, int ldlp, void* tok,
                          void* workspace, float temperature, float top_p, float min_p, int top_k, unsigned seed,
                          const void* step_ptr, void* stream) {
  if (temperature < 0.f) return VLM_ERR_ARG;
  vlm_sampler_params sp{};
  sp.temperature = temperature;
  sp.top_p = top_p;
  sp.min_p = min_p;
  sp.min_tokens_to_keep = 1;
  sp.top_k = top_k;
  sp.typical_p = 1.0;
  sp.seed = seed;
  return vlm_sample_ex(logits, ld, B, V, logprobs, scratch, ldlp, tok, workspace, &sp, step_ptr, stream);
}

/* greedy tail of the decode step in two launches: vlm_sample (temperature 0) + vlm_decode_advance + the next step's
 * vlm_embed_gather (see logprob_argmax_tail_kernel) */
extern "C" int vlm_sample_greedy_advance(const void* logits, int ld, int B, int V, void* logprobs, int ldlp, void* tok,
                                         void* workspace, void* ctx, void* pos, void* out_ring, int ring_len, void* step,
                                         const void* embed, void* h, int D, int ldh, void* stream) {
  if (!logits || !tok || !workspace || !ctx || !pos || !step || !embed || !h || B <= 0 || V <= 0) return VLM_ERR_ARG;
  if (B > TAIL_MAX_B || D % 8 || ldh % 8) return VLM_ERR_SHAPE;
  if (out_ring && ring_len <= 0) return VLM_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  unsigned* ticket = (unsigned*)workspace;
  float* ws = (float*)((char*)workspace + 256);
  float* cand_v = ws + ROW_CV;
  int* cand_i = (int*)(ws + ROW_CI);
  hipLaunchKernelGGL(lse_partial_kernel, dim3(NBLK, B), dim3(256), 0, st, (const bf16_t*)logits, ld, V, ws);
  VLM_CHECK_LAUNCH();
  hipLaunchKernelGGL(logprob_argmax_tail_kernel, dim3(NBLK, B), dim3(256), 0, st, (const bf16_t*)logits, ld, V, ws,
                     (bf16_t*)logprobs, ldlp, cand_v, cand_i, ticket, (int*)tok, (int*)ctx, (int*)pos, (int*)out_ring, ring_len,
                     (int*)step, (const bf16_t*)embed, (bf16_t*)h, D, ldh, B);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}

extern "C" int vlm_apply_logit_penalties(void* logits, int ld, int B, int V, const void* push_tok, const vlm_penalty_args* p,
                                         void* stream) {
  if (!logits || !p || !p->hist || !p->hist_len || B <= 0 || V <= 0) return VLM_ERR_ARG;
  if (p->hist_cap <= 0 || p->hist_cap > 1024 || p->rep_ctx > p->hist_cap || p->pres_ctx > p->hist_cap ||
      p->freq_ctx > p->hist_cap)
    return VLM_ERR_SHAPE;
  if (p->n_bias > 0 && (!p->bias_idx || !p->bias_val)) return VLM_ERR_ARG;
  if (p->row_params && p->bias_stride > 0 && (!p->bias_idx || !p->bias_val)) return VLM_ERR_ARG;
  if (p->row_params && p->bias_stride < 0) return VLM_ERR_ARG;
  PenaltyK k{(int*)p->hist, (int*)p->hist_len, p->hist_cap, p->rep_penalty, p->rep_ctx, p->pres_penalty, p->pres_ctx,
             p->freq_penalty, p->freq_ctx, (const int*)p->bias_idx, (const float*)p->bias_val, p->n_bias,
             (const float*)p->row_params, p->bias_stride};
  hipLaunchKernelGGL(logit_penalties_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, (bf16_t*)logits, ld, V,
                     (const int*)push_tok, k);
  VLM_CHECK_LAUNCH();
  return VLM_OK;
}
