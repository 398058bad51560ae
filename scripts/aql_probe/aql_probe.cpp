// AQL fence probe (round 5): what does a dependent chain of small kernels cost per link when the packets carry agent-scope
// acquire/release fences (what the HIP runtime puts on the nodes of a captured graph, profiles/r05_runtime_knobs_ab.txt), and
// what when they carry none and the kernels themselves use agent-scope accesses for the few words they hand over?
// Own HSA queue, own packets; barrier bit set on every packet in all variants (the chain is dependent).
//   build: g++ -O2 -I/opt/rocm/include -o aql_probe aql_probe.cpp -L/opt/rocm/lib -lhsa-runtime64
//   run:   ./aql_probe probe.hsaco
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(s_, &m); \
  fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m ? m : "?"); exit(2); } } while (0)

static hsa_agent_t g_gpu, g_cpu;
static hsa_amd_memory_pool_t g_dev_pool, g_kernarg_pool;
static bool have_gpu = false, have_cpu = false;

static hsa_status_t on_agent(hsa_agent_t a, void*) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_dev_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t fl = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  bool alloc = false;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED)) { g_dev_pool = p; return HSA_STATUS_INFO_BREAK; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_cpu_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t fl = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT)) { g_kernarg_pool = p; return HSA_STATUS_INFO_BREAK; }
  return HSA_STATUS_SUCCESS;
}

struct Kernel { uint64_t object; uint32_t kernarg, group, priv; };
static Kernel get_kernel(hsa_executable_t exe, const char* name) {
  hsa_executable_symbol_t sym;
  CHECK(hsa_executable_get_symbol_by_name(exe, (std::string(name) + ".kd").c_str(), &g_gpu, &sym));
  Kernel k;
  CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
  CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
  CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
  CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
  return k;
}

static hsa_queue_t* g_q;
static void dispatch(const Kernel& k, void* kernarg, uint32_t n_wg, int acq, int rel, hsa_signal_t done, int barrier = 1) {
  const uint64_t idx = hsa_queue_add_write_index_relaxed(g_q, 1);
  while (idx - hsa_queue_load_read_index_scacquire(g_q) >= g_q->size) { }
  hsa_kernel_dispatch_packet_t* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(g_q->base_address) + (idx & (g_q->size - 1));
  p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
  p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
  p->grid_size_x = n_wg * 256; p->grid_size_y = 1; p->grid_size_z = 1;
  p->private_segment_size = k.priv; p->group_segment_size = k.group;
  p->kernel_object = k.object; p->kernarg_address = kernarg; p->reserved2 = 0; p->completion_signal = done;
  const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (barrier << HSA_PACKET_HEADER_BARRIER) |
                          (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
  __atomic_store_n(reinterpret_cast<uint16_t*>(p), header, __ATOMIC_RELEASE);
  hsa_signal_store_screlease(g_q->doorbell_signal, idx);
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "probe.hsaco";
  CHECK(hsa_init());
  CHECK(hsa_iterate_agents(on_agent, nullptr));
  if (!have_gpu || !have_cpu) { fprintf(stderr, "no gpu/cpu agent\n"); return 2; }
  hsa_amd_agent_iterate_memory_pools(g_gpu, on_dev_pool, nullptr);
  hsa_amd_agent_iterate_memory_pools(g_cpu, on_cpu_pool, nullptr);
  CHECK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_MULTI, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &g_q));

  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); return 2; }
  fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> blob(sz);
  if (fread(blob.data(), 1, sz, f) != (size_t)sz) return 2;
  fclose(f);
  hsa_code_object_reader_t reader;
  CHECK(hsa_code_object_reader_create_from_memory(blob.data(), sz, &reader));
  hsa_executable_t exe;
  CHECK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  CHECK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
  CHECK(hsa_executable_freeze(exe, nullptr));
  const Kernel k_empty = get_kernel(exe, "k_empty"), k_plain = get_kernel(exe, "k_chain_plain"), k_agent = get_kernel(exe, "k_chain_agent"),
               k_gemv = get_kernel(exe, "k_gemv_agent");

  const int N = 1536, ROWS = 2048, COLS32 = 768;        // the gemv link: 2048 x 1536 bf16 = 6.3 MB of weights (the qkv GEMV of the headline)
  float *a, *b;
  uint32_t* w;
  CHECK(hsa_amd_memory_pool_allocate(g_dev_pool, 4096 * 4, 0, (void**)&a));
  CHECK(hsa_amd_memory_pool_allocate(g_dev_pool, 4096 * 4, 0, (void**)&b));
  CHECK(hsa_amd_memory_pool_allocate(g_dev_pool, (size_t)ROWS * COLS32 * 4 * 28, 0, (void**)&w));      // 28 distinct weight sets (> L2)
  char* ka;
  CHECK(hsa_amd_memory_pool_allocate(g_kernarg_pool, 1 << 16, 0, (void**)&ka));
  CHECK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ka));
  std::vector<float> ha(N), hb(N);
  hsa_signal_t done;
  CHECK(hsa_signal_create(1, 0, nullptr, &done));

  struct ChainArgs { const float* in; float* out; int n; };
  struct GemvArgs { const uint32_t* w; const float* x; float* y; int rows, cols32; };
  ChainArgs* ca = reinterpret_cast<ChainArgs*>(ka);
  ca[0] = {a, b, N}; ca[1] = {b, a, N};
  GemvArgs* ga = reinterpret_cast<GemvArgs*>(ka + 4096);
  for (int i = 0; i < 56; ++i) ga[i] = {w + (size_t)(i % 28) * ROWS * COLS32, (i & 1) ? b : a, (i & 1) ? a : b, ROWS, COLS32};

  const int LINKS = 2000;
  auto run = [&](const char* name, const Kernel& k, int n_wg, int acq, int rel, int mode) {
    // mode 0: no args; 1: chain ping-pong (checked); 2: gemv ping-pong
    double best = 1e30;
    int bad = -1;
    for (int rep = 0; rep < 5; ++rep) {
      if (mode == 1) {
        for (int i = 0; i < N; ++i) { ha[i] = 0.f; hb[i] = -1.f; }
        CHECK(hsa_memory_copy(a, ha.data(), N * 4));
        CHECK(hsa_memory_copy(b, hb.data(), N * 4));
      }
      hsa_signal_store_relaxed(done, 1);
      // a leading system-scope packet so that the host's initialisation is visible and the timing starts from an idle queue
      dispatch(k_empty, nullptr, 1, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM, hsa_signal_t{0});
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < LINKS; ++i) {
        void* arg = mode == 0 ? nullptr : mode == 1 ? (void*)&ca[i & 1] : (void*)&ga[i % 56];
        const bool last = i == LINKS - 1;
        dispatch(k, arg, n_wg, last ? HSA_FENCE_SCOPE_SYSTEM : acq, last ? HSA_FENCE_SCOPE_SYSTEM : rel, last ? done : hsa_signal_t{0});
      }
      while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) { }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      best = us < best ? us : best;
      if (mode == 1) {
        CHECK(hsa_memory_copy(ha.data(), a, N * 4));
        bad = 0;
        for (int i = 0; i < N; ++i) bad += ha[i] != (float)LINKS;
      }
    }
    printf("%-36s wg %4d  acquire %d release %d : %7.3f us per link", name, n_wg, acq, rel, best / LINKS);
    if (mode == 1) printf("   chain %s (%d of %d wrong)", bad ? "BROKEN" : "ok", bad, N);
    printf("\n");
    fflush(stdout);
  };
  const int A = HSA_FENCE_SCOPE_AGENT, Z = HSA_FENCE_SCOPE_NONE, S = HSA_FENCE_SCOPE_SYSTEM;
  for (int wg : {1, 256, 2048}) {
    run("empty", k_empty, wg, S, S, 0);
    run("empty", k_empty, wg, A, A, 0);
    run("empty", k_empty, wg, A, Z, 0);
    run("empty", k_empty, wg, Z, A, 0);
    run("empty", k_empty, wg, Z, Z, 0);
  }
  run("chain, plain loads/stores", k_plain, 256, A, A, 1);
  run("chain, plain loads/stores", k_plain, 256, Z, Z, 1);
  run("chain, agent-scope loads/stores", k_agent, 256, A, A, 1);
  run("chain, agent-scope loads/stores", k_agent, 256, Z, Z, 1);
  run("gemv 6.3 MB, agent-scope x / y", k_gemv, 512, S, S, 2);
  run("gemv 6.3 MB, agent-scope x / y", k_gemv, 512, A, A, 2);
  run("gemv 6.3 MB, agent-scope x / y", k_gemv, 512, Z, Z, 2);

  // ---- realistic links (probe_kernels.hip: k_link_barrier / k_link_flag): kernel arguments in device memory, as HIP places them
  {
    const Kernel k_bar = get_kernel(exe, "k_link_barrier"), k_flag = get_kernel(exe, "k_link_flag");
    struct LinkArgs { const uint32_t* w; const uint32_t* x; uint32_t* y; int* flag; int rows; int wait_for; int* err; int link; int n_wg;
                      const uint32_t* pf; long long pf_n16; int pf_wgs; int pad; };
    const size_t POOL = (size_t)1536 << 20;
    uint32_t* wp;
    CHECK(hsa_amd_memory_pool_allocate(g_dev_pool, POOL, 0, (void**)&wp));
    int* flag;
    CHECK(hsa_amd_memory_pool_allocate(g_dev_pool, 1 << 16, 0, (void**)&flag));
    const int MAXL = 2000;
    LinkArgs* dargs;
    CHECK(hsa_amd_memory_pool_allocate(g_dev_pool, sizeof(LinkArgs) * MAXL, 0, (void**)&dargs));
    std::vector<LinkArgs> h(MAXL);
    auto chain = [&](const char* name, const std::vector<int>& rows_of_layer, int layers, int mode, std::vector<double> pf_mb = {}, int pf_wgs = 256) {
      // mode 0: barrier bit + agent fences (what a HIP graph does); 1: barrier bit, no fences; 2: no barrier, no fences, flags
      const int per = (int)rows_of_layer.size(), L = per * layers;
      size_t cur = 0;
      int arrived = 0;
      for (int i = 0; i < L; ++i) {
        const int rows = rows_of_layer[i % per];
        const size_t bytes = (size_t)rows * 3072;
        if (cur + bytes > POOL) cur = 0;
        h[i] = {wp + cur / 4, (uint32_t*)((i & 1) ? b : a), (uint32_t*)((i & 1) ? a : b), flag, rows, i - 1, flag + 16, i, (rows + 7) / 8,
                nullptr, 0, 0, 0};
        cur += bytes;
        arrived += (rows + 7) / 8;
      }
      if (!pf_mb.empty()) {
        // link j of a layer prefetches pf_mb[j] MB of what the links AFTER the small ones read (in order: the bytes of link 3, then 4)
        for (int l = 0; l < layers; ++l) {
          const char* big0 = (const char*)h[l * per + 3].w;
          const size_t big_bytes = (size_t)(rows_of_layer[3] + rows_of_layer[4]) * 3072;
          const bool contiguous = (const char*)h[l * per + 4].w == big0 + (size_t)rows_of_layer[3] * 3072;
          size_t off = 0;
          for (int j = 0; j < 3 && contiguous; ++j) {
            size_t nb = (size_t)(pf_mb[j] * 1e6) / 16 * 16;
            if (off + nb > big_bytes) nb = big_bytes - off;
            if (!nb) continue;
            h[l * per + j].pf = (const uint32_t*)(big0 + off);
            h[l * per + j].pf_n16 = nb / 16;
            h[l * per + j].pf_wgs = pf_wgs;
            off += nb;
          }
        }
      }
      CHECK(hsa_memory_copy(dargs, h.data(), sizeof(LinkArgs) * L));
      double best = 1e30;
      int err = 0;
      for (int rep = 0; rep < 4; ++rep) {
        std::vector<int> zero(1 << 14, -7);          // no link number matches; the arrival counters (second half) start at 0
        for (int i = 32 + 32 * 64; i < (1 << 14); ++i) zero[i] = 0;
        zero[16] = 0;
        CHECK(hsa_memory_copy(flag, zero.data(), 1 << 16));
        hsa_signal_store_relaxed(done, 1);
        dispatch(k_empty, nullptr, 1, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM, hsa_signal_t{0});
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < L; ++i) {
          const bool last = i == L - 1;
          const int sc = mode == 0 ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE;
          dispatch(mode == 2 ? k_flag : k_bar, dargs + i, (h[i].rows + 7) / 8 + h[i].pf_wgs, last ? HSA_FENCE_SCOPE_SYSTEM : sc, last ? HSA_FENCE_SCOPE_SYSTEM : sc,
                   last ? done : hsa_signal_t{0}, (mode == 2 && !last) ? 0 : 1);
        }
        while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) { }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        best = us < best ? us : best;
        int back[32];
        CHECK(hsa_memory_copy(back, flag, sizeof(back)));
        err |= back[16];
      }
      size_t bytes_layer = 0;
      for (int r : rows_of_layer) bytes_layer += (size_t)r * 3072;
      printf("%-38s %-30s : %8.3f us per layer (%5.2f us per link, %5.2f TB/s)%s\n", name,
             mode == 0 ? "barrier bit + agent fences" : mode == 1 ? "barrier bit, no fences" : "no barrier, arrival counters", best / layers,
             best / L, bytes_layer * layers / best * 1e-6, err ? "   PROTOCOL ERROR" : "");
      fflush(stdout);
    };
    for (int mode = 0; mode < 3; ++mode) chain("qkv-sized links (6.3 MB)", {2048}, 560, mode);
    for (int mode = 0; mode < 3; ++mode) chain("o-sized links (4.7 MB)", {1536}, 560, mode);
    for (int mode = 0; mode < 3; ++mode) chain("gate/up-sized links (55 MB)", {17920}, 280, mode);
    // a layer's five launches by weight bytes: qkv 6.3, attention stand-in 0.8, o 4.7, gate/up 55, down 27.5 MB
    for (int mode = 0; mode < 3; ++mode) chain("layer mix (94 MB, 5 links)", {2048, 256, 1536, 17920, 8960}, 112, mode);
    // the three latency-bound launches of a layer carry prefetch workgroups for the two bandwidth-bound ones (82.6 MB)
    const std::vector<int> LAYER = {2048, 256, 1536, 17920, 8960};
    chain("mix + prefetch 3 x 6 MB", LAYER, 112, 0, {6, 6, 6});
    chain("mix + prefetch 3 x 12 MB", LAYER, 112, 0, {12, 12, 12});
    chain("mix + prefetch 3 x 18.4 MB (gate/up)", LAYER, 112, 0, {18.4, 18.4, 18.4});
    chain("mix + prefetch 3 x 27.6 MB (all)", LAYER, 112, 0, {27.6, 27.6, 27.6});
    chain("mix + prefetch 3 x 18.4 MB, 512 wgs", LAYER, 112, 0, {18.4, 18.4, 18.4}, 512);
    chain("mix + prefetch 3 x 18.4 MB, 128 wgs", LAYER, 112, 0, {18.4, 18.4, 18.4}, 128);
  }
  hsa_shut_down();
  return 0;
}
