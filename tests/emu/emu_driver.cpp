// tests/emu/emu_driver.cpp — runs the solver body (csrc/kas_solver_body.h) on CPU fibers.
// TEST INFRASTRUCTURE: see tests/emu/kas_wave.h.  Exposes kas_emu_solve_batch() with the
// semantics of kas_solve_host(), including the product's own planning code (kas_plan_math.h).
#include <string.h>

#include <string>
#include <vector>

#include "emu/kas_wave.h"     // defines KAS_WAVE_H_ first, so the body's own #include "kas_wave.h" is a no-op
#include "kas_solver_body.h"

namespace kasw {

Emu g_emu;
static const size_t STACK_BYTES = 256 * 1024;
static char* g_stacks = nullptr;

struct Tramp { void (*fn)(void*); void* arg; };
static Tramp g_tramp;

static void lane_entry() {
  g_tramp.fn(g_tramp.arg);
  g_emu.done[g_emu.cur_lane] = true;
  swapcontext(&g_emu.lane_ctx[g_emu.cur_lane], &g_emu.main_ctx);
}

int run_wave(void (*fn)(void*), void* arg) {
  Emu& e = g_emu;
  if (!g_stacks) g_stacks = (char*)malloc(64 * STACK_BYTES);
  g_tramp.fn = fn; g_tramp.arg = arg;
  for (int i = 0; i < 64; ++i) {
    e.done[i] = false; e.kind[i] = K_NONE;
    getcontext(&e.lane_ctx[i]);
    e.lane_ctx[i].uc_stack.ss_sp = g_stacks + (size_t)i * STACK_BYTES;
    e.lane_ctx[i].uc_stack.ss_size = STACK_BYTES;
    e.lane_ctx[i].uc_link = &e.main_ctx;
    makecontext(&e.lane_ctx[i], lane_entry, 0);
  }
  for (;;) {
    int alive = 0;
    for (int i = 0; i < 64; ++i) {
      if (e.done[i]) continue;
      e.cur_lane = i;
      swapcontext(&e.main_ctx, &e.lane_ctx[i]);
      if (!e.done[i]) ++alive;
    }
    if (alive == 0) return 0;
    // all surviving lanes are parked: they must be at the same kind of collective, and no
    // lane may have exited while others still wait
    int k = -1;
    for (int i = 0; i < 64; ++i) {
      if (e.done[i]) { fprintf(stderr, "emu: lane %d exited while others wait at a collective\n", i); return -1; }
      if (k < 0) k = e.kind[i];
      else if (e.kind[i] != k) {
        fprintf(stderr, "emu: wave divergence: lane %d at collective kind %d, lane 0.. at %d\n", i, e.kind[i], k);
        return -1;
      }
    }
    e.collectives++;
  }
}

}  // namespace kasw

namespace {

struct RunArgs { const KasLaunch* a; int32_t s; unsigned char* lds; int W; };

template <int W> void run_one(void* p) {
  RunArgs* r = (RunArgs*)p;
  kas::solve_scenario<W>(*r->a, r->s, r->lds);
}

}  // namespace

extern "C" __attribute__((visibility("default")))
int kas_emu_solve_batch(const kas_batch_desc* b, const kas_tables* t, unsigned flags, char* errbuf, int errlen) {
  KasShape sh;
  std::string err;
  int rc = kas_shape_batch(b, &sh, &err);
  if (rc != KAS_E_OK) {
    if (errbuf && errlen > 0) { strncpy(errbuf, err.c_str(), (size_t)errlen - 1); errbuf[errlen - 1] = 0; }
    return rc;
  }
  std::vector<uint64_t> accmask((size_t)sh.accmask_words + 1, 0xDEADBEEFDEADBEEFull);
  std::vector<unsigned char> lds((size_t)sh.lds.total + 64, 0xCD);
  KasLaunch a;
  a.scen = b->scenarios; a.topics = b->topics; a.node_id = b->node_id; a.node_rack = b->node_rack;
  a.cur = t->cur; a.out = t->out; a.aux = t->aux; a.ctx = t->ctx;
  a.topic_results = t->topic_results; a.scenario_results = t->scenario_results;
  a.accmask = accmask.data(); a.accmask_off = sh.accmask_off.data(); a.stats = nullptr;
  a.n_scenarios = b->n_scenarios; a.n_max = sh.n_max; a.idmap_entries = sh.idmap_entries;
  a.need_bsearch = sh.need_bsearch; a.hist_separate = sh.hist_separate; a.flags = flags;
  for (int32_t s = 0; s < b->n_scenarios; ++s) {
    memset(lds.data(), 0xCD, lds.size());   // LDS is uninitialised on hardware too
    RunArgs ra{&a, s, lds.data(), sh.W};
    void (*fn)(void*) = nullptr;
    switch (sh.Wc) {                       // the same width classes the product launcher uses
      case 2: fn = run_one<2>; break;
      case 3: fn = run_one<3>; break;
      case 4: fn = run_one<4>; break;
      case 5: fn = run_one<5>; break;
      default: fn = run_one<8>; break;
    }
    if ((size_t)sh.lds.total + 64 > lds.size()) lds.resize((size_t)sh.lds.total + 64, 0xCD);
    ra.lds = lds.data();
    if (kasw::run_wave(fn, &ra) != 0) {
      if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "wave divergence in scenario %d", s);
      return -100;
    }
  }
  return KAS_E_OK;
}

extern "C" __attribute__((visibility("default")))
long kas_emu_collectives(void) { return kasw::g_emu.collectives; }
