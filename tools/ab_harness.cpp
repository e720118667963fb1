// tools/ab_harness.cpp — A/B of builds of the C-ABI library without Python: the same seeded batch, resident in HBM,
// through every library named on the command line (dlopen, one after the other); per library the fill / order
// kernel durations (kas_plan_phase_times_us: HIP events on the launch stream), the plan's kernel string, and a
// checksum of the result records, which must be the same for all of them.  Seconds from process start to result
// (no torch import): made for the short end of a GPU budget.  TEST / MEASUREMENT TOOLING, not a product path.
//   hipcc -O2 -std=c++17 -I include -o tools/ab_harness tools/ab_harness.cpp -ldl
//   tools/ab_harness c3 SCENARIOS REPS lib.so [lib2.so ...]    100k partitions x 1k brokers x 20 racks x RF 3, remove 1 broker
//   tools/ab_harness c3mix SCENARIOS REPS lib.so [...]         the same tables; per scenario remove 1 / remove <= 5 / add <= 50 / both
//   tools/ab_harness c5 1 REPS lib.so [...]                    1M x 5k x 40 racks x RF 5, remove every 50th + add 200
//   tools/ab_harness c5norack 1 REPS lib.so [...]              the same with every broker its own rack (--disable_rack_awareness)
//   tools/ab_harness shape:P:N:R:RF SCENARIOS REPS lib.so [...]   any shape, remove 1 broker
//   tools/ab_harness multi:P:N:R SCENARIOS REPS lib.so [...]      three topics per scenario ("topic-0..2": lists 3, 2 and 3 wide, P rows
//                                                                 each): topic changes and rows narrower than the batch in the order kernels
//   AB_INFLIGHT=K:STEPS:REPEATS  in addition: K plans on K streams (own out tables, the same cur), STEPS solves round-robin
//                                between two synchronisations, REPEATS times: scenarios/s by the host clock (bench.py's regime)
//   AB_FLAGS=n                   kas_plan_set_flags(n) on every plan (KAS_PLAN_* of include/kas_abi.h)
//   AB_DISTINCT=1                in flight: every slot its own copy of the cur table (as bench.py's slots have)
//   AB_CELLS16=1                 the batch as uint16 node-index cells through kas_plan_create16 / kas_solve_device16 (ABI v5)
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kas_abi.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint32_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); }
  uint32_t below(uint32_t n) { return (uint32_t)(((uint64_t)next() * n) >> 32); }
};

struct Api {
  void* h = nullptr;
  int (*ctx_create)(int, kas_ctx**);
  void (*ctx_destroy)(kas_ctx*);
  int (*plan_create)(kas_ctx*, const kas_batch_desc*, kas_plan**);
  void (*plan_destroy)(kas_plan*);
  int (*plan_describe)(const kas_plan*, char*, int);
  int (*solve_device)(kas_plan*, const kas_tables*, void*);
  int (*ctx_synchronize)(kas_ctx*);
  int (*phase_times)(kas_plan*, double*, double*, int*);
  const char* (*last_error)(void);
  int (*set_flags)(kas_plan*, unsigned);
  int (*plan_create16)(kas_ctx*, const kas_batch_desc*, kas_plan**) = nullptr;      // (ABI v5; absent from older builds)
  int (*solve_device16)(kas_plan*, const kas_tables16*, void*) = nullptr;
  bool cells16 = false;                                     // AB_CELLS16: plans and solves on 16-bit cells
  int make_plan(kas_ctx* c, const kas_batch_desc* b, kas_plan** p) { return cells16 ? plan_create16(c, b, p) : plan_create(c, b, p); }
  int solve(kas_plan* p, const kas_tables* t, void* st) {   // (kas_tables16 has kas_tables' layout: pointer types apart)
    return cells16 ? solve_device16(p, reinterpret_cast<const kas_tables16*>(t), st) : solve_device(p, t, st);
  }
  bool load(const char* path) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return false; }
#define SYM(field, name) field = (decltype(field))dlsym(h, name); if (!field) { fprintf(stderr, "%s: no %s\n", path, name); return false; }
    SYM(ctx_create, "kas_ctx_create") SYM(ctx_destroy, "kas_ctx_destroy") SYM(plan_create, "kas_plan_create")
    SYM(plan_destroy, "kas_plan_destroy") SYM(plan_describe, "kas_plan_describe") SYM(solve_device, "kas_solve_device")
    SYM(ctx_synchronize, "kas_ctx_synchronize") SYM(phase_times, "kas_plan_phase_times_us") SYM(last_error, "kas_last_error")
    SYM(set_flags, "kas_plan_set_flags")
#undef SYM
    plan_create16 = (decltype(plan_create16))dlsym(h, "kas_plan_create16");
    solve_device16 = (decltype(solve_device16))dlsym(h, "kas_solve_device16");
    cells16 = getenv("AB_CELLS16") != nullptr;
    if (cells16 && (!plan_create16 || !solve_device16)) { fprintf(stderr, "%s: no kas_plan_create16 / kas_solve_device16\n", path); return false; }
    return true;
  }
};

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s c3|c5 SCENARIOS REPS lib.so [lib.so ...]\n", argv[0]); return 1; }
  const auto t_start = std::chrono::steady_clock::now();
  const std::string mode = argv[1];
  const bool c5 = mode == "c5" || mode == "c5norack", mix = mode == "c3mix", norack = mode == "c5norack";
  const int S = atoi(argv[2]), reps = atoi(argv[3]);
  int32_t P = c5 ? 1000000 : 100000, N0 = c5 ? 5000 : 1000, R = c5 ? 40 : 20, RF = c5 ? 5 : 3;
  if (mode.rfind("shape:", 0) == 0 && sscanf(mode.c_str(), "shape:%d:%d:%d:%d", &P, &N0, &R, &RF) != 4) { fprintf(stderr, "shape:P:N:R:RF\n"); return 1; }
  // ---- the batch (host): G-like start — RF distinct racks per partition, one broker inside each — and the action
  const bool multi = mode.rfind("multi:", 0) == 0;
  if (multi) { RF = 3; if (sscanf(mode.c_str(), "multi:%d:%d:%d", &P, &N0, &R) != 3) { fprintf(stderr, "multi:P:N:R\n"); return 1; } }
  const int T = multi ? 3 : 1;
  const int32_t t_rf[3] = {RF, multi ? 2 : RF, RF};
  const int32_t t_hash[3] = {multi ? -1139260654 : 3644, -1139260653, -1139260652};
  int64_t cells_per_scen = 0;
  for (int k = 0; k < T; ++k) cells_per_scen += (int64_t)P * t_rf[k];
  std::vector<int32_t> cur((size_t)S * cells_per_scen), node_id, node_rack;
  std::vector<kas_scenario_desc> scen((size_t)S);
  std::vector<kas_topic_desc> topics((size_t)S * T);
  for (int s = 0; s < S; ++s) {
    Rng g(1000 + s);
    const uint32_t per_rack = (uint32_t)(N0 / R);
    int64_t toff = (int64_t)s * cells_per_scen;
    for (int k = 0; k < T; ++k) {
      const int32_t rf = t_rf[k];
      int32_t* c = cur.data() + toff;
      for (int32_t p = 0; p < P; ++p) {
        int32_t racks[8];
        for (int r = 0; r < rf; ++r) {
          for (;;) {
            const int32_t kk = (int32_t)g.below((uint32_t)R);
            bool dup = false;
            for (int q = 0; q < r; ++q) dup = dup || racks[q] == kk;
            if (!dup) { racks[r] = kk; break; }
          }
          c[(size_t)p * rf + r] = racks[r] + R * (int32_t)g.below(per_rack);      // broker b sits on rack b mod R
        }
      }
      kas_topic_desc td;
      memset(&td, 0, sizeof td);
      td.name_hash = t_hash[k]; td.n_partitions = P; td.cur_width = rf; td.rf = rf; td.out_width = rf;
      td.cur_off = toff; td.out_off = toff;
      td.cur_len_off = -1; td.in_partitions_off = -1; td.part_id_off = -1;
      topics[(size_t)s * T + k] = td;
      toff += (int64_t)P * rf;
    }
    const int64_t off = (int64_t)node_id.size();
    int32_t n = 0;
    // the scenario's broker set: which of 0..N0-1 go, how many of N0, N0 + 1, ... come
    const int kind = mix ? s % 4 : 0;
    const int n_gone = c5 ? 0 : (kind == 0 ? 1 : (kind == 2 ? 0 : 1 + s % 5));
    const int n_new = c5 ? 200 : ((kind == 2 || kind == 3) ? (s == 2 ? 50 : 1 + (s * 7) % 50) : 0);
    std::vector<char> gone_b((size_t)N0, 0);
    for (int j = 0; j < n_gone; ++j) gone_b[(size_t)(((int64_t)s * 37 + 11 + (int64_t)j * 97) % N0)] = 1;
    for (int32_t b = 0; b < N0 + n_new; ++b) {
      const bool removed = b < N0 && (c5 ? b % 50 == 0 : gone_b[(size_t)b] != 0);
      if (removed) continue;
      node_id.push_back(b); node_rack.push_back(norack ? n : b % R); ++n;
    }
    scen[s] = kas_scenario_desc{n, s * T, T, 0, off, -1};
  }
  kas_batch_desc bd;
  bd.n_scenarios = S; bd.n_topics = S * T; bd.scenarios = scen.data(); bd.topics = topics.data();
  bd.node_id = node_id.data(); bd.node_rack = node_rack.data(); bd.node_pool_len = (int64_t)node_id.size();
  if (getenv("AB_CELLS16")) {
    // 16-bit cells (kas_plan_create16 / kas_solve_device16): every topic's rows as uint16 node indices — packed at the start
    // of the pools, the descriptors' offsets count cells either way; node i has id i
    for (int s = 0; s < S; ++s) {
      const kas_scenario_desc& sd = scen[s];
      std::vector<int32_t> index_of((size_t)N0 + 64, -1);
      for (int32_t i = 0; i < sd.n_nodes; ++i) index_of[(size_t)node_id[(size_t)(sd.node_off + i)]] = i;
      for (int k = 0; k < T; ++k) {
        const kas_topic_desc& td = topics[(size_t)s * T + k];
        const int32_t* src = cur.data() + td.cur_off;
        uint16_t* dst = reinterpret_cast<uint16_t*>(cur.data()) + td.cur_off;   // (in place, ascending: byte 2 i <= byte 4 i)
        const int64_t n = (int64_t)td.n_partitions * td.cur_width;
        for (int64_t i = 0; i < n; ++i) { const int32_t ix = index_of[(size_t)src[i]]; dst[i] = ix < 0 ? (uint16_t)0xffffu : (uint16_t)ix; }
      }
      for (int32_t i = 0; i < sd.n_nodes; ++i) node_id[(size_t)(sd.node_off + i)] = i;
    }
  }
  const auto t_gen = std::chrono::steady_clock::now();
  if (const char* dump = getenv("AB_DUMP")) {
    // the batch as raw int32 arrays for a checker outside this tool (tests/test_ab_harness.py: the oracle on the same tables):
    // header {S, T, P, RF of topic 0..2}, per scenario {n_nodes}, node_id pool, node_rack pool, cur pool
    FILE* f = fopen(dump, "wb");
    if (!f) { perror(dump); return 3; }
    const int32_t head[6] = {S, T, P, t_rf[0], t_rf[1], t_rf[2]};
    fwrite(head, 4, 6, f);
    for (int s = 0; s < S; ++s) fwrite(&scen[s].n_nodes, 4, 1, f);
    fwrite(node_id.data(), 4, node_id.size(), f);
    fwrite(node_rack.data(), 4, node_rack.size(), f);
    fwrite(cur.data(), 4, cur.size(), f);
    fclose(f);
  }
  if (const char* emu = getenv("AB_EMU")) {
    // no GPU at hand: the same batch through the CPU emulator of the kernel source (tests/emu/libkas_emu.so) — checks
    // the descriptors this tool builds, nothing else
    void* h = dlopen(emu, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", emu, dlerror()); return 3; }
    auto solve = (int (*)(const kas_batch_desc*, const kas_tables*, unsigned, char*, int))dlsym(h, "kas_emu_solve_batch");
    if (!solve) { fprintf(stderr, "%s: no kas_emu_solve_batch\n", emu); return 3; }
    std::vector<int32_t> out(cur.size(), -1);
    std::vector<kas_topic_result> tr((size_t)S * T);
    std::vector<kas_scenario_result> sr((size_t)S);
    kas_tables ht;
    memset(&ht, 0, sizeof ht);
    ht.cur = cur.data(); ht.out = out.data(); ht.topic_results = tr.data(); ht.scenario_results = sr.data();
    ht.cur_len = ht.out_len = (int64_t)cur.size();
    char err[512] = "";
    const int rc = solve(&bd, &ht, 0u, err, (int)sizeof err);
    uint64_t sum = 0;
    for (const kas_scenario_result& r : sr)
      sum += r.digest * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)r.status * 1315423911ull + (uint64_t)(uint32_t)r.moved_replicas;
    printf("emulator rc %d %s records %016llx\n", rc, err, (unsigned long long)sum);
    for (int s = 0; s < S && (s < 4 || getenv("AB_DUMP")); ++s)
      printf("  scenario %d: status %d moved_replicas %d moved_partitions %d digest %016llx\n", s, sr[s].status, sr[s].moved_replicas,
             sr[s].moved_partitions, (unsigned long long)sr[s].digest);
    return rc;
  }
  // ---- tables in HBM
  int32_t *d_cur = nullptr, *d_out = nullptr;
  kas_topic_result* d_tr = nullptr;
  kas_scenario_result* d_sr = nullptr;
  const size_t cells = cur.size();
  HIP_OK(hipMalloc(&d_cur, 4 * cells)); HIP_OK(hipMalloc(&d_out, 4 * cells));
  HIP_OK(hipMalloc(&d_tr, sizeof(kas_topic_result) * S * T)); HIP_OK(hipMalloc(&d_sr, sizeof(kas_scenario_result) * S));
  HIP_OK(hipMemcpy(d_cur, cur.data(), 4 * cells, hipMemcpyHostToDevice));
  kas_tables t;
  memset(&t, 0, sizeof t);
  t.cur = d_cur; t.out = d_out; t.topic_results = d_tr; t.scenario_results = d_sr;
  t.cur_len = (int64_t)cells; t.out_len = (int64_t)cells;
  printf("%s: %d scenario(s) of %d partitions x %d brokers x RF %d, %d timed solves per library; generated in %.2f s\n", argv[1], S, P,
         N0, RF, reps, std::chrono::duration<double>(t_gen - t_start).count());
  uint64_t first_sum = 0;
  bool all_same = true;
  for (int li = 4; li < argc; ++li) {
    Api api;
    if (!api.load(argv[li])) return 3;
    kas_ctx* ctx = nullptr;
    kas_plan* plan = nullptr;
    if (api.ctx_create(0, &ctx) != 0) { fprintf(stderr, "%s: kas_ctx_create: %s\n", argv[li], api.last_error()); return 4; }
    if (api.make_plan(ctx, &bd, &plan) != 0) { fprintf(stderr, "%s: kas_plan_create: %s\n", argv[li], api.last_error()); return 4; }
    const unsigned flags = getenv("AB_FLAGS") ? (unsigned)strtoul(getenv("AB_FLAGS"), nullptr, 0) : 0u;
    if (flags && api.set_flags(plan, flags) != 0) { fprintf(stderr, "%s: kas_plan_set_flags: %s\n", argv[li], api.last_error()); return 4; }
    char what[1024];
    api.plan_describe(plan, what, sizeof what);
    HIP_OK(hipMemset(d_out, 0xff, 4 * cells)); HIP_OK(hipMemset(d_sr, 0, sizeof(kas_scenario_result) * S));
    if (api.solve(plan, &t, nullptr) != 0 || api.ctx_synchronize(ctx) != 0) { fprintf(stderr, "%s: solve: %s\n", argv[li], api.last_error()); return 5; }
    double f = 0, o = 0;
    int n = 0;
    api.phase_times(plan, &f, &o, &n);                       // (resets the accumulator: the first solve is warm-up)
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
      if (api.solve(plan, &t, nullptr) != 0) { fprintf(stderr, "%s: solve: %s\n", argv[li], api.last_error()); return 5; }
    if (api.ctx_synchronize(ctx) != 0) { fprintf(stderr, "%s: sync: %s\n", argv[li], api.last_error()); return 5; }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    api.phase_times(plan, &f, &o, &n);
    std::vector<kas_scenario_result> sr((size_t)S);
    HIP_OK(hipMemcpy(sr.data(), d_sr, sizeof(kas_scenario_result) * S, hipMemcpyDeviceToHost));
    uint64_t sum = 0;
    int ok = 0;
    int64_t moved = 0;
    for (const kas_scenario_result& r : sr) {
      sum += r.digest * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)r.status * 1315423911ull + (uint64_t)(uint32_t)r.moved_replicas;
      ok += r.status == KAS_OK; moved += r.moved_replicas;
    }
    if (li == 4) first_sum = sum; else all_same = all_same && sum == first_sum;
    printf("%-44s fill %9.1f us  order %9.1f us  (%d launches, %.3f ms per solve by the host clock)  ok %d/%d moved %lld  records %016llx\n   %s\n",
           argv[li], f, o, n, wall_ms, ok, S, (long long)moved, (unsigned long long)sum, what);
    if (const char* inf = getenv("AB_INFLIGHT")) {
      int K = 8, steps = 20, repeats = 3;
      sscanf(inf, "%d:%d:%d", &K, &steps, &repeats);
      std::vector<kas_plan*> plans((size_t)K);
      std::vector<hipStream_t> streams((size_t)K);
      std::vector<kas_tables> tabs((size_t)K, t);
      for (int k = 0; k < K; ++k) {
        if (api.make_plan(ctx, &bd, &plans[k]) != 0) { fprintf(stderr, "%s: kas_plan_create: %s\n", argv[li], api.last_error()); return 4; }
        if (flags) api.set_flags(plans[k], flags);
        HIP_OK(hipStreamCreateWithFlags(&streams[k], hipStreamNonBlocking));
        if (getenv("AB_DISTINCT") && k > 0) {                    // every slot its own copy of the cur table (bench.py's regime: no slot
          int32_t* c = nullptr;                                  // finds another slot's rows in a cache)
          HIP_OK(hipMalloc(&c, 4 * cells)); HIP_OK(hipMemcpy(c, d_cur, 4 * cells, hipMemcpyDeviceToDevice));
          tabs[k].cur = c;
        }
        int32_t* o = nullptr; kas_topic_result* tr = nullptr; kas_scenario_result* srk = nullptr;
        HIP_OK(hipMalloc(&o, 4 * cells)); HIP_OK(hipMalloc(&tr, sizeof(kas_topic_result) * S * T)); HIP_OK(hipMalloc(&srk, sizeof(kas_scenario_result) * S));
        tabs[k].out = o; tabs[k].topic_results = tr; tabs[k].scenario_results = srk;
        if (api.solve(plans[k], &tabs[k], streams[k]) != 0) { fprintf(stderr, "solve: %s\n", api.last_error()); return 5; }   // set-up solve
      }
      HIP_OK(hipDeviceSynchronize());
      printf("   in flight, %d plans x %d steps:", K, steps);
      for (int rep = 0; rep < repeats; ++rep) {
        const auto a0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i)
          if (api.solve(plans[i % K], &tabs[i % K], streams[i % K]) != 0) { fprintf(stderr, "solve: %s\n", api.last_error()); return 5; }
        HIP_OK(hipDeviceSynchronize());
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - a0).count();
        printf(" %.1fk scenarios/s", (double)S * steps / sec / 1e3);
      }
      // every slot's records must be the batch's
      bool slots_ok = true;
      for (int k = 0; k < K; ++k) {
        std::vector<kas_scenario_result> srk((size_t)S);
        HIP_OK(hipMemcpy(srk.data(), tabs[k].scenario_results, sizeof(kas_scenario_result) * S, hipMemcpyDeviceToHost));
        uint64_t sk = 0;
        for (const kas_scenario_result& r : srk)
          sk += r.digest * 0x9E3779B97F4A7C15ull + (uint64_t)(uint32_t)r.status * 1315423911ull + (uint64_t)(uint32_t)r.moved_replicas;
        slots_ok = slots_ok && sk == sum;
        api.plan_destroy(plans[k]);
        HIP_OK(hipStreamDestroy(streams[k]));
        if (tabs[k].cur != d_cur) HIP_OK(hipFree((void*)tabs[k].cur));
        HIP_OK(hipFree(tabs[k].out)); HIP_OK(hipFree(tabs[k].topic_results)); HIP_OK(hipFree(tabs[k].scenario_results));
      }
      printf("  (every slot's records %s)\n", slots_ok ? "equal the batch's" : "DIFFER");
      all_same = all_same && slots_ok;
    }
    api.plan_destroy(plan);
    api.ctx_destroy(ctx);
  }
  printf("%s; total %.1f s\n", all_same ? "records identical across the libraries" : "RECORDS DIFFER",
         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
  return all_same ? 0 : 6;
}
