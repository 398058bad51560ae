// Standalone A/B driver for the decode step on MI355X: links libvlm_hip.so (the C ABI only), builds a
// Qwen2-VL-2B-shaped language model from synthetic weights with hipMalloc (no torch: the process starts in
// milliseconds, which matters when GPU time is budgeted by the minute) and times graph replays of
// vlm_llm_decode_graph_launch under different vlm_llm_set_tuning / vlm_decode_args.flags settings.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 scripts/decode_probe.cpp -Iinclude -Lmlx-vlm_amd/lib -lvlm_hip \
//         -Wl,-rpath,'$ORIGIN/../mlx-vlm_amd/lib' -o scripts/bin/decode_probe
//   scripts/bin/decode_probe [--steps 300] [--ctx 450] [--variant flags,psplit,gv,am ...] [--micro] [--no-hot]
//
// Every variant restarts from the same device state and must reproduce the baseline's tokens bit for bit
// (prefetch and the fused tail change scheduling only).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "vlm_hip.h"

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e__ = (x);                                                                  \
    if (e__ != hipSuccess) {                                                               \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
      exit(2);                                                                             \
    }                                                                                      \
  } while (0)
#define RC(x)                                                              \
  do {                                                                     \
    int r__ = (x);                                                         \
    if (r__ != 0) {                                                        \
      fprintf(stderr, "vlm rc=%d at %s:%d\n", r__, __FILE__, __LINE__);    \
      exit(3);                                                             \
    }                                                                      \
  } while (0)

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)i * 0x9E3779B1u + seed;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    const float u = ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f) - 0.5f;   // (-0.5, 0.5)
    const float v = 2.0f * scale * u;
    uint32_t b;
    memcpy(&b, &v, 4);
    b += 0x7fffu + ((b >> 16) & 1u);
    p[i] = (uint16_t)(b >> 16);
  }
}
__global__ void fill_const_bf16(uint16_t* p, size_t n, uint16_t v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void touch_kernel(const uint4* p, size_t n16, unsigned* never) {   // plain-load sweep (cache warmer)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned s = 0;
  for (; i < n16; i += stride) { uint4 v = p[i]; s ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (s == 0x12345u && never) never[0] = s;
}

struct Dev {
  std::vector<void*> owned;
  template <typename T>
  T* alloc(size_t n, bool zero = true) {
    void* p;
    CK(hipMalloc(&p, n * sizeof(T)));
    if (zero) CK(hipMemset(p, 0, n * sizeof(T)));
    owned.push_back(p);
    return (T*)p;
  }
  uint16_t* rnd(size_t n, uint32_t seed, float scale) {
    uint16_t* p = alloc<uint16_t>(n, false);
    fill_bf16<<<2048, 256>>>(p, n, seed, scale);
    return p;
  }
  uint16_t* ones(size_t n) {
    uint16_t* p = alloc<uint16_t>(n, false);
    fill_const_bf16<<<(unsigned)((n + 255) / 256), 256>>>(p, n, 0x3f80);
    return p;
  }
};

struct ModelDims {
  int hidden = 1536, layers = 28, inter = 8960, heads = 12, kv_heads = 2, head_dim = 128, vocab = 151936;
};

struct Model {
  ModelDims d;
  void* h = nullptr;
  std::vector<vlm_llm_layer> L;
  vlm_llm_globals g{};
  vlm_kv_pool kv{};
  int max_pages = 256;
};

static Model build_model(Dev& dev, const ModelDims& d, const Model* share) {
  Model m;
  m.d = d;
  vlm_llm_config cfg{d.hidden, d.layers, d.inter, d.heads, d.kv_heads, d.head_dim, d.vocab, 1e-6f, 16, 24, 0.f};
  RC(vlm_llm_create(&cfg, &m.h));
  const size_t D = d.hidden, QKV = (size_t)(d.heads + 2 * d.kv_heads) * d.head_dim;
  for (int i = 0; i < d.layers; ++i) {
    vlm_llm_layer w{};
    if (share) {
      w = share->L[i];
    } else {
      w.ln1_w = dev.ones(D);
      w.wqkv = dev.rnd(QKV * D, 11 + 7 * i, 0.035f);
      w.bqkv = dev.rnd(QKV, 12 + 7 * i, 0.02f);
      w.wo = dev.rnd(D * d.heads * d.head_dim, 13 + 7 * i, 0.035f);
      w.ln2_w = dev.ones(D);
      w.wgu = dev.rnd(2 * (size_t)d.inter * D, 14 + 7 * i, 0.035f);
      w.wdown = dev.rnd((size_t)d.inter * D, 15 + 7 * i, 0.035f);
    }
    m.L.push_back(w);
    RC(vlm_llm_set_layer(m.h, i, &w));
  }
  if (share && share->d.vocab == d.vocab) {
    m.g = share->g;
  } else {
    m.g.embed = dev.rnd((size_t)d.vocab * D, 5, 0.05f);
    m.g.lm_head = m.g.embed;
    m.g.final_norm_w = dev.ones(D);
    float inv[64];
    for (int j = 0; j < 64; ++j) inv[j] = 1.0f / powf(1e6f, (float)(2 * j) / 128.0f);
    float* dinv = dev.alloc<float>(64);
    CK(hipMemcpy(dinv, inv, sizeof(inv), hipMemcpyHostToDevice));
    m.g.inv_freq = dinv;
  }
  RC(vlm_llm_set_globals(m.h, &m.g));
  if (share) {
    m.kv = share->kv;
  } else {
    const size_t per_layer = (size_t)m.max_pages * d.kv_heads * d.head_dim * 64;   // elements, one sequence row
    m.kv.kpool = dev.rnd(per_layer * d.layers, 21, 0.5f);
    m.kv.vpool = dev.rnd(per_layer * d.layers, 22, 0.5f);
    m.kv.layer_stride = per_layer;
    m.kv.block_table = nullptr;   // identity layout
    m.kv.max_pages = m.max_pages;
  }
  RC(vlm_llm_set_kv(m.h, &m.kv));
  return m;
}

struct State {
  vlm_decode_args a{};
  int* ring = nullptr;
  int ring_len = 64;
};

static State make_state(Dev& dev, const ModelDims& d, int B) {
  State s;
  vlm_decode_args& a = s.a;
  const size_t QKV = (size_t)(d.heads + 2 * d.kv_heads) * d.head_dim;
  a.B = B;
  a.tok = dev.alloc<int>(B);
  a.pos = dev.alloc<int>(B);
  a.ctx = dev.alloc<int>(B);
  a.step = dev.alloc<int>(1);
  a.h = dev.alloc<uint16_t>((size_t)B * d.hidden);
  a.qkv = dev.alloc<uint16_t>(B * QKV);
  a.attn = dev.alloc<uint16_t>((size_t)B * d.heads * d.head_dim);
  a.act = dev.alloc<uint16_t>((size_t)B * d.inter);
  a.logits = dev.alloc<uint16_t>((size_t)B * d.vocab);
  a.logprobs = nullptr;
  a.scratch = dev.alloc<uint16_t>((size_t)B * d.vocab);
  a.part_o = dev.alloc<float>((size_t)B * d.heads * 32 * d.head_dim);
  a.part_ml = dev.alloc<float>((size_t)B * d.heads * 32 * 2);
  a.sample_ws = dev.alloc<char>(vlm_sample_workspace_bytes(B));
  s.ring = dev.alloc<int>((size_t)s.ring_len * B);
  a.out_ring = s.ring;
  a.ring_len = s.ring_len;
  a.nsplit = 1;
  a.temperature = 0.f;
  a.top_p = 1.f;
  a.min_p = 0.f;
  a.top_k = 0;
  a.seed = 0;
  a.flags = 0;
  return s;
}

static void reset_state(const Model& m, State& s, int ctx0, hipStream_t st) {
  const int B = s.a.B;
  std::vector<int> tok(B), pos(B), ctx(B);
  for (int b = 0; b < B; ++b) { tok[b] = 1000 + 17 * b; pos[b] = ctx0 + 3; ctx[b] = ctx0; }
  int zero = 0;
  CK(hipMemcpy(s.a.tok, tok.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(s.a.pos, pos.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(s.a.ctx, ctx.data(), B * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(s.a.step, &zero, 4, hipMemcpyHostToDevice));
  CK(hipMemset(s.ring, 0xff, (size_t)s.ring_len * B * 4));
  // h = embed[tok] (what the caller of a fused-tail step provides once)
  RC(vlm_embed_gather(s.a.tok, m.g.embed, s.a.h, B, m.d.hidden, m.d.hidden, m.d.vocab, st));
  CK(hipStreamSynchronize(st));
}

struct Variant {
  int flags = 1, psplit = 16, gv = 1, am = 1;     // fused greedy tail, page-split width, GEMV variant bits, merge in o_proj
  std::string name() const {
    char b[160];
    snprintf(b, sizeof b, "flags=%d psplit=%d gv=%d am=%d", flags, psplit, gv, am);
    return b;
  }
};

// -> microseconds per step; tokens of the first ring_len steps in `toks`
static double run_variant(const Model& m, State& s, const Variant& v, int ctx0, int warm, int steps, std::vector<int>* toks,
                          hipStream_t st) {
  RC(vlm_llm_set_tuning(m.h, VLM_TUNE_ATTN_PAGESPLIT, v.psplit));
  RC(vlm_llm_set_tuning(m.h, VLM_TUNE_GEMV_VARIANT, v.gv));
  RC(vlm_llm_set_tuning(m.h, VLM_TUNE_ATTN_MERGE, v.am));
  s.a.flags = v.flags;
  RC(vlm_llm_set_kv(m.h, &m.kv));
  reset_state(m, s, ctx0, st);
  RC(vlm_llm_decode_graph_build(m.h, &s.a, st));
  // first ring_len steps: token record
  for (int i = 0; i < s.ring_len; ++i) RC(vlm_llm_decode_graph_launch(m.h, st));
  CK(hipStreamSynchronize(st));
  if (toks) {
    toks->resize((size_t)s.ring_len * s.a.B);
    CK(hipMemcpy(toks->data(), s.ring, toks->size() * 4, hipMemcpyDeviceToHost));
  }
  reset_state(m, s, ctx0, st);
  for (int i = 0; i < warm; ++i) RC(vlm_llm_decode_graph_launch(m.h, st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < steps; ++i) RC(vlm_llm_decode_graph_launch(m.h, st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return (double)ms * 1e3 / steps;
}

static double time_loop(hipStream_t st, int reps, const std::function<void()>& fn) {
  fn();
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) fn();
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return (double)ms * 1e3 / reps;
}

static void micro(Dev& dev, const Model& m, hipStream_t st) {
  const ModelDims& d = m.d;
  const int D = d.hidden, I = d.inter, NL = d.layers;
  uint16_t* x = dev.rnd(D, 91, 1.0f);
  uint16_t* act = dev.rnd(I, 92, 1.0f);
  uint16_t* y = dev.alloc<uint16_t>(2 * (size_t)I);
  uint16_t* hres = dev.alloc<uint16_t>(D);
  struct Case { const char* name; size_t bytes; std::function<void(int)> launch; };
  std::vector<Case> cases = {
      {"gate_up", (size_t)2 * 2 * I * D,
       [&](int l) { RC(vlm_gemv_bf16(x, m.L[l].wgu, nullptr, nullptr, m.L[l].ln2_w, y, 1, 2 * I, D, D, D, I, 0, 1e-6f, VLM_EPI_SWIGLU, st)); }},
      {"down", (size_t)2 * I * D,
       [&](int l) { RC(vlm_gemv_bf16(act, m.L[l].wdown, nullptr, hres, nullptr, hres, 1, D, I, I, I, D, D, 0.f, VLM_EPI_RESIDUAL, st)); }},
      {"o_proj", (size_t)2 * D * D,
       [&](int l) { RC(vlm_gemv_bf16(x, m.L[l].wo, nullptr, hres, nullptr, hres, 1, D, D, D, D, D, D, 0.f, VLM_EPI_RESIDUAL, st)); }},
  };
  printf("\n== micro: GEMV cold (28 layers cycled, > Infinity Cache) vs hot (one layer repeated) vs warmed by a plain-load sweep\n");
  for (auto& c : cases) {
    const double cold = time_loop(st, 10, [&] { for (int l = 0; l < NL; ++l) c.launch(l); }) / NL;
    const double hot = time_loop(st, 10, [&] { for (int l = 0; l < NL; ++l) c.launch(3); }) / NL;
    printf("  %-8s %9zu B  cold %6.2f us (%5.2f TB/s)   hot %6.2f us (%5.2f TB/s)\n", c.name, c.bytes, cold,
           c.bytes / cold * 1e-6, hot, c.bytes / hot * 1e-6);
  }
  // a plain-load sweep of layer l+1's gate/up weights, then the GEMV on them: does the sweep's data survive in MALL?
  {
    const size_t bytes = (size_t)2 * 2 * I * D;
    const double sweep = time_loop(st, 5, [&] { for (int l = 0; l < NL; ++l) touch_kernel<<<1024, 256, 0, st>>>((const uint4*)m.L[l].wgu, bytes / 16, nullptr); }) / NL;
    const double both = time_loop(st, 5, [&] {
      for (int l = 0; l < NL; ++l) {
        touch_kernel<<<1024, 256, 0, st>>>((const uint4*)m.L[l].wgu, bytes / 16, nullptr);
        cases[0].launch(l);
      }
    }) / NL;
    printf("  sweep(gate_up) alone %6.2f us (%5.2f TB/s); sweep + GEMV %6.2f us => GEMV after sweep %6.2f us\n", sweep,
           bytes / sweep * 1e-6, both, both - sweep);
  }
}

int main(int argc, char** argv) {
  int steps = 300, warm = 30, ctx0 = 450, B = 1;
  bool do_micro = false, hot_layer = true, use_table = false;
  std::vector<Variant> variants;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--steps" && i + 1 < argc) steps = atoi(argv[++i]);
    else if (a == "--ctx" && i + 1 < argc) ctx0 = atoi(argv[++i]);
    else if (a == "--batch" && i + 1 < argc) B = atoi(argv[++i]);
    else if (a == "--micro") do_micro = true;
    else if (a == "--no-hot") hot_layer = false;
    else if (a == "--block-table") use_table = true;
    else if (a == "--variant" && i + 1 < argc) {
      Variant v;
      sscanf(argv[++i], "%d,%d,%i,%d", &v.flags, &v.psplit, &v.gv, &v.am);
      variants.push_back(v);
    }
  }
  if (variants.empty()) {
    variants = {
        {1, 16, 1, 1},   // the product's step
        {0, 16, 1, 1},   // without the fused greedy tail
        {1, 16, 0, 1},   // down projection at 4 rows per workgroup
        {1, 16, 1, 0},   // merge by the attention launch's last arriver
        {1, 16, 1, 1},
    };
  }
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  Dev dev;
  ModelDims d;
  Model m = build_model(dev, d, nullptr);
  if (use_table) {   // paged form of the same placement: row b owns pages [b * max_pages, (b + 1) * max_pages)
    std::vector<int> tab((size_t)8 * m.max_pages);
    for (size_t i = 0; i < tab.size(); ++i) tab[i] = (int)i;
    int* dt = dev.alloc<int>(tab.size());
    CK(hipMemcpy(dt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    m.kv.block_table = dt;
  }
  CK(hipDeviceSynchronize());
  printf("abi v%d, model 2B dims, ctx %d, batch %d, %d timed steps\n", vlm_abi_version(), ctx0, B, steps);
  State s = make_state(dev, d, B);
  std::vector<int> base_toks;
  const double lm_bytes = 2.0 * (28.0 * 46797824 + 1536 + 233373696);
  for (size_t vi = 0; vi < variants.size(); ++vi) {
    std::vector<int> toks;
    const double us = run_variant(m, s, variants[vi], ctx0, warm, steps, &toks, st);
    if (vi == 0) base_toks = toks;
    const bool same = toks == base_toks;
    int first_diff = -1;
    for (size_t i = 0; i < toks.size() && i < base_toks.size(); ++i)
      if (toks[i] != base_toks[i]) { first_diff = (int)i; break; }
    if (!same) printf("   first differing token at step %d of %zu\n", first_diff, toks.size());
    const double bytes = lm_bytes + 28672.0 * (ctx0 + steps / 2);
    if (getenv("VLM_ATTN_STAMPS")) {   // timeline of the LAST attention launch (library built with -DVLM_ATTN_TIMELINE)
      float st32[32];
      CK(hipMemcpy(st32, s.a.part_o, sizeof(st32), hipMemcpyDeviceToHost));
      const char* names[9] = {"start", "q", "kv-issued", "QK", "softmax", "PV", "pre-bar", "post-bar", "end"};
      for (int w = 0; w < 2; ++w) {
        printf("   attention wave %d:", w ? 7 : 0);
        for (int i = 1; i < 9; ++i) printf(" %s=%.2f", names[i], st32[16 * w + i]);
        printf("\n");
      }
    }
    printf("%-56s %8.1f us/step  %7.1f tok/s  %5.3f of 8 TB/s  launches %3d  tokens %s\n", variants[vi].name().c_str(), us,
           1e6 / us * B, bytes / us * 1e-6 / 8.0, vlm_llm_decode_launches(m.h), same ? "== baseline" : "DIFFER");
    fflush(stdout);
  }
  printf("first tokens:");
  for (int i = 0; i < 12 && i < (int)base_toks.size(); ++i) printf(" %d", base_toks[i]);
  printf("\n");
  if (hot_layer) {
    // Infinity-Cache-resident bound: 1- and 2-layer models over the SAME weights with a tiny vocabulary (one layer =
    // 93.6 MB, two = 187 MB < 256 MB): the difference is the time of one layer whose weights are on die
    ModelDims d1 = d, d2 = d;
    d1.layers = 1; d2.layers = 2; d1.vocab = d2.vocab = 2048;
    Model m1 = build_model(dev, d1, &m);
    Model m2 = build_model(dev, d2, &m);
    m2.g = m1.g;
    RC(vlm_llm_set_globals(m2.h, &m2.g));
    State s1 = make_state(dev, d1, B);
    Variant v0{1, 16, 1, 1};
    const double t1 = run_variant(m1, s1, v0, ctx0, warm, steps, nullptr, st);
    const double t2 = run_variant(m2, s1, v0, ctx0, warm, steps, nullptr, st);
    printf("\n== cache-resident bound: 1 layer %.1f us/step, 2 layers %.1f us/step -> one on-die layer %.1f us (HBM-cold: see baseline / 28)\n",
           t1, t2, t2 - t1);
  }
  if (do_micro) micro(dev, m, st);
  return 0;
}
