// tests/emu/emu_driver.cpp — runs the solver body (csrc/kas_solver_body.h) on CPU fibers.
// TEST INFRASTRUCTURE: see tests/emu/kas_wave.h.  Exposes kas_emu_solve_batch() with the
// semantics of kas_solve_host(), including the product's own planning code (kas_plan_math.h).
#include <string.h>

#include <string>
#include <cstdlib>
#include <vector>

#include "emu/kas_wave.h"     // defines KAS_WAVE_H_ first, so the body's own #include "kas_wave.h" is a no-op
#include "kas_solver_body.h"

namespace kasw {

Emu g_emu;
static const size_t STACK_BYTES = 192 * 1024;
static char* g_stacks = nullptr;

struct Tramp { void (*fn)(void*); void* arg; };
static Tramp g_tramp;

static void lane_entry() {
  g_tramp.fn(g_tramp.arg);
  g_emu.state[g_emu.cur] = S_DONE;
  kas_emu_switch(&g_emu.lane_ctx[g_emu.cur], &g_emu.main_ctx);
  abort();                                                   // a finished fiber is never resumed
}

#if KAS_EMU_FAST_SWITCH
// kas_emu_switch(from, to): push the callee-saved registers, park the stack pointer in *from, take
// *to's, pop, return into the other fiber.  (MXCSR / x87 control words are the process defaults in
// every fiber and nothing here changes them.)
asm(".text\n"
    ".globl kas_emu_switch\n"
    ".hidden kas_emu_switch\n"
    ".type kas_emu_switch,@function\n"
    "kas_emu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq (%rsi), %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size kas_emu_switch, .-kas_emu_switch\n");

// a fresh fiber: six zeroed registers, then lane_entry as the address the first switch returns to,
// entered with the stack the ABI promises a function (rsp + 8 a multiple of 16)
static void make_fiber(Ctx* c, char* stack, size_t bytes) {
  uintptr_t top = ((uintptr_t)stack + bytes) & ~(uintptr_t)15;
  void** sp = (void**)(top - 64);
  for (int i = 0; i < 6; ++i) sp[i] = nullptr;
  sp[6] = (void*)&lane_entry;
  sp[7] = nullptr;
  c->sp = sp;
}
#else
static void make_fiber(Ctx* c, char* stack, size_t bytes) {
  getcontext(&c->uc);
  c->uc.uc_stack.ss_sp = stack;
  c->uc.uc_stack.ss_size = bytes;
  c->uc.uc_link = &g_emu.main_ctx.uc;
  makecontext(&c->uc, lane_entry, 0);
}
#endif

// KAS_EMU_CHAOS=<seed>: waves no longer advance in step.  Each round a wave whose lanes have all
// arrived at a collective is released only with probability 1/2 (never none of them), and now and
// then one wave is held back for a long stretch — relative wave speeds on hardware are arbitrary,
// and the protocols between waves (ring tags, P4 progress words) must not depend on them.
static uint64_t g_chaos = 0;
static uint32_t chaos_next() {
  g_chaos ^= g_chaos << 13; g_chaos ^= g_chaos >> 7; g_chaos ^= g_chaos << 17;
  return (uint32_t)(g_chaos >> 32);
}

long g_last_block_rounds = 0;   // scheduling rounds of the last run_block (KAS_EMU_STATS)

int run_block(void (*fn)(void*), void* arg, int n_waves) {
  Emu& e = g_emu;
  static int chaos_init = 0;
  if (!chaos_init) {
    chaos_init = 1;
    const char* c = getenv("KAS_EMU_CHAOS");
    if (c && *c) g_chaos = 0x9E3779B97F4A7C15ull * (uint64_t)(strtoull(c, nullptr, 10) + 1);
  }
  // KAS_EMU_WAVE_DIV="w:k[,w:k...]": wave w is released only every k-th scheduling round — a crude way
  // of making one wavefront the slow one (on hardware the class-1 solver of the wide order kernel is the
  // bottleneck and always has a full hand; with every wave at the same speed it is starved instead)
  static int wave_div[KAS_EMU_MAX_LANES / 64];
  static int div_init = 0;
  if (!div_init) {
    div_init = 1;
    for (int w = 0; w < KAS_EMU_MAX_LANES / 64; ++w) wave_div[w] = 1;
    const char* c = getenv("KAS_EMU_WAVE_DIV");
    while (c && *c) {
      char* end = nullptr;
      const long w = strtol(c, &end, 10);
      if (!end || *end != ':') break;
      const long k = strtol(end + 1, &end, 10);
      if (w >= 0 && w < KAS_EMU_MAX_LANES / 64 && k >= 1) wave_div[w] = (int)k;
      c = (*end == ',') ? end + 1 : nullptr;
    }
  }
  int held_wave = -1;
  long held_rounds = 0, rounds = 0;
  const int n = 64 * n_waves;
  if (n > KAS_EMU_MAX_LANES) return -1;
  if (!g_stacks) g_stacks = (char*)malloc((size_t)KAS_EMU_MAX_LANES * STACK_BYTES);
  g_tramp.fn = fn; g_tramp.arg = arg;
  e.n_lanes = n;
  for (int i = 0; i < n; ++i) {
    e.state[i] = S_RUNNABLE; e.kind[i] = K_NONE;
    make_fiber(&e.lane_ctx[i], g_stacks + (size_t)i * STACK_BYTES, STACK_BYTES);
  }
  for (;;) {
    // 1. run every runnable fiber until it parks at a collective or finishes
    int ran = 0, done = 0;
    for (int i = 0; i < n; ++i) {
      if (e.state[i] == S_DONE) { ++done; continue; }
      if (e.state[i] != S_RUNNABLE) continue;
      e.cur = i;
      kas_emu_switch(&e.main_ctx, &e.lane_ctx[i]);
      ++ran;
      if (e.state[i] == S_DONE) ++done;
    }
    if (done == n) { g_last_block_rounds = rounds; return 0; }
    // 2. release every wave whose 64 fibers all wait at the same wave collective; a workgroup
    //    barrier releases when every fiber of the block waits at it
    int released = 0, at_sync = 0, n_ready = 0;
    int ready_waves[KAS_EMU_MAX_LANES / 64];
    for (int w = 0; w < n_waves; ++w) {
      int parked = 0, finished = 0, k = -1;
      bool mixed = false;
      for (int i = 64 * w; i < 64 * w + 64; ++i) {
        if (e.state[i] == S_DONE) { ++finished; continue; }
        if (e.state[i] == S_PARKED) {
          ++parked;
          if (k < 0) k = e.kind[i]; else if (e.kind[i] != k) mixed = true;
        }
      }
      if (finished == 64) continue;
      if (parked + finished < 64) continue;                // cannot happen after step 1
      if (finished > 0) { fprintf(stderr, "emu: wave %d: %d lanes exited while others wait at a collective\n", w, finished); return -1; }
      if (mixed) {
        fprintf(stderr, "emu: wave %d: divergence, lanes wait at different collectives:", w);
        for (int i = 64 * w; i < 64 * w + 64; ++i) fprintf(stderr, " %d", e.kind[i]);
        fprintf(stderr, "\n");
        return -1;
      }
      if (k == K_SYNC) { at_sync += 64; continue; }
      ready_waves[n_ready++] = w;
    }
    if (n_ready > 0) {
      int first_released = -1;
      if (g_chaos != 0 && n_waves > 1) {
        if (held_rounds > 0) --held_rounds; else held_wave = -1;
        if (held_wave < 0 && (chaos_next() & 1023u) == 0) { held_wave = (int)(chaos_next() % (uint32_t)n_waves); held_rounds = 50 + (long)(chaos_next() % 2000u); }
      }
      for (int r = 0; r < n_ready; ++r) {
        const int w = ready_waves[r];
        bool go = true;
        if (g_chaos != 0 && n_waves > 1) go = w != held_wave && (chaos_next() & 1u) != 0;
        if (wave_div[w] > 1 && n_waves > 1 && (rounds % wave_div[w]) != 0) go = false;
        if (!go) continue;
        for (int i = 64 * w; i < 64 * w + 64; ++i) e.state[i] = S_RUNNABLE;
        ++released; e.collectives++;
        if (first_released < 0) first_released = w;
      }
      if (released == 0) {                                 // never stall everybody
        int w = ready_waves[chaos_next() % (uint32_t)n_ready];
        if (w == held_wave && n_ready > 1) w = ready_waves[(w == ready_waves[0]) ? 1 : 0];
        for (int i = 64 * w; i < 64 * w + 64; ++i) e.state[i] = S_RUNNABLE;
        ++released; e.collectives++;
      }
    }
    if (++rounds > 400000000L) { fprintf(stderr, "emu: no end in sight after %ld rounds (livelock?)\n", rounds); return -1; }
    if (at_sync == n) {
      for (int i = 0; i < n; ++i) e.state[i] = S_RUNNABLE;
      ++released;
    } else if (at_sync > 0 && at_sync == n - done && done > 0) {
      fprintf(stderr, "emu: some waves exited while others wait at the workgroup barrier\n");
      return -1;
    }
    if (released == 0 && ran == 0) { fprintf(stderr, "emu: deadlock (nothing runnable, nothing released)\n"); return -1; }
    if (released == 0) {
      // every live fiber is parked and no group is complete
      bool any_runnable = false;
      for (int i = 0; i < n; ++i) any_runnable = any_runnable || e.state[i] == S_RUNNABLE;
      if (!any_runnable) { fprintf(stderr, "emu: deadlock at a collective\n"); return -1; }
    }
  }
}

}  // namespace kasw

namespace {

struct RunArgs { const KasLaunch* a; int32_t s; unsigned char* lds; };

template <int W, int M32C = 0> void run_fill_slim(void* p) {
  RunArgs* r = (RunArgs*)p;
  kas::fill_scenario<W, 4, true, M32C>(*r->a, r->s, r->lds);
}
template <int W, int NW> void run_fill(void* p) {
  RunArgs* r = (RunArgs*)p;
  kas::fill_scenario<W, NW>(*r->a, r->s, r->lds);
}
template <int W, int M32C = 0> void run_p4(void* p) {
  RunArgs* r = (RunArgs*)p;
  kas::p4_scenario<W, 1, false, M32C>(*r->a, r->s, r->lds);   // (one wavefront, as kas_p4_kernel is launched; its instance per mid-row layout)
}
template <int W, int G, bool PK> void run_order_tickets(void* p) {
  RunArgs* r = (RunArgs*)p;
  if constexpr (W <= 3) kas::order_tickets<W, G, PK>(*r->a, r->s, r->lds);
}
void run_permutation(void* p) {
  RunArgs* r = (RunArgs*)p;
  kas::order_permutation(*r->a, r->lds);
}
template <int W> void run_order_relax_wide(void* p) {
  RunArgs* r = (RunArgs*)p;
  if constexpr (W == 4 || W == 5) kas::order_relax_wide<W>(*r->a, r->s, r->lds);
}
template <int W> void run_order_wide(void* p) {
  RunArgs* r = (RunArgs*)p;
  kas::order_tickets_wide<W>(*r->a, r->s, r->lds);
}
template <int W, bool DUAL, bool CTX, bool VERIFY = false, bool C16 = false, bool IDL = false> void run_order_relax(void* p) {
  RunArgs* r = (RunArgs*)p;
  if constexpr (W <= 3) kas::order_relax<W, DUAL, CTX, VERIFY, C16, IDL>(*r->a, r->s, r->lds);
}
template <int W, bool DUAL, bool C16, bool IDL, bool M32 = false, bool QUAD = false> void run_p4_order(void* p) {
  RunArgs* r = (RunArgs*)p;
  if constexpr (W <= 3) kas::p4_order_scenario<W, DUAL, C16, IDL, M32, QUAD>(*r->a, r->s, r->lds);
}
// the instances for dword mid rows (KAS_FLAG_MID32; kas_order_relax_m32_pick in kas_hip.hip)
template <bool DUAL, bool QUAD = false> void run_order_relax_m32(void* p) {
  RunArgs* r = (RunArgs*)p;
  kas::order_relax<3, DUAL, false, false, false, true, false, true, QUAD>(*r->a, r->s, r->lds);
}
typedef void (*relax_fn)(void*);
template <bool VERIFY, bool C16, bool IDL> relax_fn relax_pick(int Wc, bool dual, bool ctx) {   // as kas_order_relax_pick (kas_hip.hip)
  if (Wc <= 2) return ctx ? run_order_relax<2, false, true, VERIFY, C16, IDL> : run_order_relax<2, false, false, VERIFY, C16, IDL>;
  if (ctx) return dual ? run_order_relax<3, true, true, VERIFY, C16, IDL> : run_order_relax<3, false, true, VERIFY, C16, IDL>;
  return dual ? run_order_relax<3, true, false, VERIFY, C16, IDL> : run_order_relax<3, false, false, VERIFY, C16, IDL>;
}
template <int W> void run_order_rounds(void* p) {
  RunArgs* r = (RunArgs*)p;
  kas::order_scenario_rounds<W>(*r->a, r->s, r->lds);
}

struct SpreadArgs { const KasLaunch* a; int32_t s, c; unsigned char* lds; };
template <int W> void run_spread_a(void* p) { SpreadArgs* r = (SpreadArgs*)p; kas::spread_pass_a<W>(*r->a, r->s, r->c, r->lds); }
template <int W> void run_spread_b(void* p) { SpreadArgs* r = (SpreadArgs*)p; kas::spread_pass_b<W>(*r->a, r->s, r->c, r->lds); }
template <int W> void run_spread_p4(void* p) { SpreadArgs* r = (SpreadArgs*)p; kas::spread_p4<W, KAS_SPREAD_P4_WAVES>(*r->a, r->s, r->lds); }
template <int W> void spread_quota_all(const KasLaunch& a) {
  for (int32_t s = 0; s < a.n_scenarios; ++s)
    for (int32_t n = 0; n < a.n_max; ++n) kas::spread_quota<W>(a, s, n);
}

// kas_fill_kernel as it is launched: workgroup `block` of `grid` runs kas::fill_block — its loop over the scenarios it takes (by index,
// or by rank among the flagged ones behind the slim kernel / the spread fill / the wide form's check), the LDS NOT cleared between them
struct BlockArgs { const KasLaunch* a; int32_t block, grid; unsigned char* lds; };
template <int W, int NW> void run_fill_block(void* p) {
  BlockArgs* r = (BlockArgs*)p;
  kas::fill_block<W, NW>(*r->a, r->block, r->grid, r->lds);
}
typedef void (*run_fn)(void*);
template <int NW> run_fn fill_for_w(int Wc) {
  switch (Wc) {                            // the same width classes the product launcher uses
    case 2: return run_fill_block<2, NW>;
    case 3: return run_fill_block<3, NW>;
    case 4: return run_fill_block<4, NW>;
    case 5: return run_fill_block<5, NW>;
    default: return run_fill_block<8, NW>;
  }
}
// workgroups of an emulated kas_fill_kernel launch: KAS_EMU_FILL_GRID (default 3: most batches of the suites then have workgroups
// that take several scenarios in a row, as a launch on fewer workgroups than scenarios does on the GPU), at most one per scenario
static int32_t emu_fill_grid(int32_t n_scenarios) {
  const char* e = getenv("KAS_EMU_FILL_GRID");
  int32_t g = e ? atoi(e) : 3;
  if (g < 1) g = 1;
  return g < n_scenarios ? g : (n_scenarios > 0 ? n_scenarios : 1);
}
static int g_last_handback = -1;           // what the last by-rank launch left in KasLaunch::handback (-1: there was none)
// one emulated launch of kas_fill_kernel; returns the workgroup that failed, or -1
static int32_t emu_launch_fill(run_fn fill, KasLaunch& a, int NW, std::vector<unsigned char>& lds, int32_t n_scenarios, bool want_handback) {
  const int32_t G = emu_fill_grid(n_scenarios);
  int32_t hb = -1;
  a.handback = want_handback ? &hb : nullptr;
  for (int32_t blk = 0; blk < G; ++blk) {
    memset(lds.data(), 0xCD, lds.size());   // LDS is uninitialised on hardware too
    BlockArgs ra{&a, blk, G, lds.data()};
    if (kasw::run_block(fill, &ra, NW) != 0) { a.handback = nullptr; return blk; }
  }
  a.handback = nullptr;
  if (want_handback) g_last_handback = hb;
  return -1;
}
template <int G, bool PK> run_fn tickets_for_g(int Wc) {
  switch (Wc) {
    case 2: return run_order_tickets<2, G, PK>;
    default: return run_order_tickets<3, G, PK>;
  }
}
run_fn rounds_for(int Wc) {
  switch (Wc) {
    case 2: return run_order_rounds<2>;
    case 3: return run_order_rounds<3>;
    case 4: return run_order_rounds<4>;
    case 5: return run_order_rounds<5>;
    default: return run_order_rounds<8>;
  }
}

}  // namespace

// rows the ticket-form solver decided inside queues during the last kas_emu_solve_batch (summed
// over scenarios): lets a CPU test assert that the queue path ran, not only the one-row path
static long g_last_queue_rows = 0;
static int g_last_p4_order = 0;       // the last solve ran first fit inside the order kernel's workgroup (kas_p4_order_kernel)
static int g_last_relax_idl = 0;      // the last relaxation-form launch read its broker ids from the LDS
static long g_last_slim_fill = 0;    // scenarios the slim fill kernel solved itself (not handed back) in the last kas_emu_solve_batch
static long g_last_relax_quad = 0;   // the last kas_emu_solve_batch ran the relaxation form over quad tiles
static long g_last_mid32 = 0;        // the last kas_emu_solve_batch moved its mid rows as one dword each (KAS_FLAG_MID32)
static long g_last_index_rows = 0;   // topics whose fill took the index rows (fill_pass_a_fused<EMIT>) in the last kas_emu_solve_batch
static int g_last_fused = 0;   // the last kas_emu_solve_batch ran the fill with per-chunk histograms
static int g_last_spread = 0;  // scenarios the spread fill solved itself (not handed back) in the last kas_emu_solve_batch
static int g_last_order_form = 0;   // 1: ticket form (lists <= 3 wide), 2: wide ticket form, 3: relaxation form, 0: round form (the last solve's plan)
static int g_last_split_p4 = 0;     // the last solve ran its first fit in kas_p4_kernel (KAS_FLAG_SPLIT_P4)
static long g_last_relax_tiles = 0, g_last_relax_evals = 0, g_last_relax_slow = 0;   // relaxation form: tiles, evaluations, tiles off the straight-line path
static int g_last_flagged = 0; // scenarios a ticket form left to the round form (Context counters too large for its fields)
// flags: low byte = KAS_FLAG_*, bits 8..11 = wavefronts per scenario of the fill kernel, bits
// 12..15 = scenarios per wavefront of the ticket-form order kernel (0 = the planner's choice),
// bit 16 = KAS_FLAG_TICKET_ORDER (the ticket form where the relaxation form would run), bits 17 / 18 = tiles of 64 rows /
// double tiles in the relaxation form whatever the batch size, bit 21 = KAS_FLAG_NO_RTN_QUOTA
static int emu_solve(const kas_batch_desc* b, const kas_tables* t, unsigned flags, char* errbuf, int errlen, bool c16);

extern "C" __attribute__((visibility("default")))
int kas_emu_solve_batch(const kas_batch_desc* b, const kas_tables* t, unsigned flags, char* errbuf, int errlen) {
  return emu_solve(b, t, flags, errbuf, errlen, false);
}

// kas_plan_create16 + kas_solve_device16 (ABI v5): t->cur / t->out point at uint16 node-index cells, b->node_id is not read
extern "C" __attribute__((visibility("default")))
int kas_emu_solve_batch16(const kas_batch_desc* b, const kas_tables* t, unsigned flags, char* errbuf, int errlen) {
  std::vector<int32_t> ids((size_t)(b->node_pool_len > 0 ? b->node_pool_len : 0), 0);
  for (int32_t s = 0; s < b->n_scenarios; ++s)
    for (int32_t i = 0; i < b->scenarios[s].n_nodes; ++i) ids[(size_t)(b->scenarios[s].node_off + i)] = i;
  kas_batch_desc ib = *b;
  ib.node_id = ids.data();
  return emu_solve(&ib, t, flags, errbuf, errlen, true);
}

static int emu_solve(const kas_batch_desc* b, const kas_tables* t, unsigned flags, char* errbuf, int errlen, bool c16) {
  KasShape sh;
  std::string err;
  int rc = kas_shape_batch(b, &sh, &err, (int)((flags >> 8) & 0xfu), (int)((flags >> 12) & 0xfu));
  if (rc != KAS_E_OK) {
    if (errbuf && errlen > 0) { strncpy(errbuf, err.c_str(), (size_t)errlen - 1); errbuf[errlen - 1] = 0; }
    return rc;
  }
  if (c16 && (sh.Wc > 3 || !(sh.relax_ok || sh.round_fits))) {   // (kas_plan_build's refusal for plans with 16-bit cells)
    if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "16-bit cells: lists up to 3 wide, relaxation or round form");
    return KAS_E_UNSUPPORTED;
  }
  if (c16 && kas_flags_want_tickets(flags) && !sh.round_fits) {   // (kas_plan_set_flags' refusal)
    if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "16-bit cells: no ticket form");
    return KAS_E_UNSUPPORTED;
  }
  // (the launch decisions of kas_launch_plan in kas_hip.hip)
  const bool relax = sh.relax_ok && !(flags & KAS_FLAG_ROUND_ORDER) && !(kas_flags_want_tickets(flags) && sh.tickets_ok);
  const bool tickets = !relax && sh.tickets_ok && !(flags & KAS_FLAG_ROUND_ORDER) && !c16;
  const bool relaxw = !relax && !c16 && sh.relaxw_ok && kas_relaxw_wanted(flags);   // relaxation form for lists 4 and 5 wide
  const bool wide = !relaxw && sh.wide_ok && !(flags & KAS_FLAG_ROUND_ORDER) && !c16;
  g_last_order_form = relax ? 3 : (relaxw ? 4 : (tickets ? 1 : (wide ? 2 : 0)));
  g_last_relax_tiles = 0; g_last_relax_evals = 0; g_last_relax_slow = 0;
  std::vector<uint64_t> accmask((size_t)sh.accmask_words + 1, 0xDEADBEEFDEADBEEFull);
  std::vector<int32_t> orph((size_t)sh.orph_ints + 64, (int32_t)0xDEADBEEF);
  const bool fused = sh.fused_ok && !(flags & KAS_FLAG_TWO_PASS_HIST) && !(flags & KAS_FLAG_GENERIC_FILL);
  size_t lds_bytes = (size_t)(fused ? sh.lds_fused.total : sh.lds.total);
  if (sh.round_fits && (size_t)kas_order_round_lds(sh.n_max, sh.Wc) > lds_bytes) lds_bytes = (size_t)kas_order_round_lds(sh.n_max, sh.Wc);
  if ((flags & KAS_FLAG_ROUND_ORDER) && !sh.round_fits) {
    if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "round form does not fit LDS");
    return KAS_E_UNSUPPORTED;
  }
  if ((size_t)kas_order_ticket_lds(sh.n_max, sh.G, 0) > lds_bytes) lds_bytes = (size_t)kas_order_ticket_lds(sh.n_max, sh.G, 0);
  if ((size_t)kas_order_relax_lds(sh.n_max, 1, 1, 1) > lds_bytes) lds_bytes = (size_t)kas_order_relax_lds(sh.n_max, 1, 1, 1);   // (an upper bound)
  if (wide && (size_t)kas_order_wide_lds(sh.n_max) > lds_bytes) lds_bytes = (size_t)kas_order_wide_lds(sh.n_max);
  if (lds_bytes < sizeof(int32_t) * (KAS_PERM_BINS + 8)) lds_bytes = sizeof(int32_t) * (KAS_PERM_BINS + 8);
  std::vector<unsigned char> lds(lds_bytes + 64, 0xCD);
  KasLaunch a;
  a.scen = b->scenarios; a.topics = b->topics; a.node_id = b->node_id; a.node_rack = b->node_rack;
  a.cur = t->cur; a.out = t->out; a.aux = t->aux; a.ctx = t->ctx;
  a.topic_results = t->topic_results; a.scenario_results = t->scenario_results;
  a.accmask = accmask.data(); a.accmask_off = sh.accmask_off.data();
  std::vector<int64_t> stats((size_t)KAS_STATS_PER_SCENARIO * (size_t)(b->n_scenarios + 1), 0);
  a.stats = stats.data();
  g_last_queue_rows = 0;
  g_last_fused = fused ? 1 : 0;
  a.orph = orph.data(); a.orph_off = sh.orph_off.data();
  std::vector<int32_t> perm((size_t)b->n_scenarios + 1, -1);
  a.perm = nullptr;
  std::vector<int32_t> ord_flag((size_t)b->n_scenarios + 1, 0);   // scenarios a ticket form leaves to the round form
  a.ord_flag = ord_flag.data();
  g_last_flagged = 0;
  a.n_scenarios = b->n_scenarios; a.n_max = sh.n_max; a.idmap_entries = sh.idmap_entries;
  a.need_bsearch = sh.need_bsearch;
  // (bit 64 of the caller's word is KAS_PLAN_NO_INDEX_ROWS, of a launch word KAS_FLAG_ONLY_FLAGGED: as kas_plan_set_flags / kas_plan_index_rows)
  const bool index_rows = !c16 && kas_index_rows_wanted(flags) && sh.Wc <= 3 && fused && !(flags & KAS_FLAG_NO_RTN_QUOTA) &&
                          sh.n_max < 0x3fff && sh.idmap_entries > 0;
  a.flags = (flags & (0xff0000ffu | KAS_FLAG_TICKET_ORDER | KAS_FLAG_RELAX_TILES_64 | KAS_FLAG_RELAX_TILES_128 | KAS_FLAG_FILL_WITH_P4 | KAS_FLAG_SPLIT_P4) & ~(KAS_FLAG_FUSED_HIST | KAS_FLAG_ONLY_FLAGGED | KAS_FLAG_ORDER_FLAGGED)) |
            (sh.with_x ? 0u : KAS_FLAG_GENERIC_FILL) | (fused ? KAS_FLAG_FUSED_HIST : 0u) |
            (kas_relax_double_tiles(flags, b->n_scenarios) ? KAS_FLAG_RELAX_DUAL : 0u) |
            ((flags & KAS_FLAG_NO_RTN_QUOTA) ? 0u : KAS_FLAG_LANE_ORDER) | (c16 ? KAS_FLAG_CELLS16 : 0u) |
            (index_rows ? KAS_FLAG_INDEX_ROWS : 0u);
  g_last_index_rows = 0;
  g_last_mid32 = 0;
  auto bad = [&](const char* what, int32_t s) {
    if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "wave divergence / deadlock in the %s kernel, scenario %d", what, s);
    return -100;
  };
  // spread fill (same decision as kas_solve_device): passes A and B over one-wavefront workgroups
  a.sp_hist = nullptr; a.sp_quota = nullptr; a.sp_node = nullptr; a.sp_flag = nullptr; a.sp_oc = nullptr; a.sp_chunks = 0;
  a.handback = nullptr;                                  // (set by emu_launch_fill for a by-rank launch)
  g_last_spread = 0;
  const int32_t CH = (!c16 && sh.NW == 4 && sh.Wc >= 3 && sh.Wc <= 5 && !(flags & KAS_FLAG_GENERIC_FILL))
                         ? kas_spread_chunks(sh, b->n_scenarios, kas_batch_single_topic(b), (flags & KAS_FLAG_SPREAD_FILL) != 0) : 0;
  std::vector<int32_t> sp_hist, sp_quota, sp_node, sp_flag, sp_oc;
  if (CH > 0) {
    const size_t S = (size_t)b->n_scenarios, NM = (size_t)sh.n_max;
    sp_hist.assign(S * (size_t)CH * (size_t)sh.Wc * NM + 1, (int32_t)0xDEADBEEF);
    sp_quota.assign(S * (size_t)CH * NM + 1, (int32_t)0xDEADBEEF);
    sp_node.assign(S * 2 * NM + 1, (int32_t)0xDEADBEEF);
    sp_flag.assign(S + 1, 0);
    sp_oc.assign(S * (size_t)(CH + 2) + 1, 0);
    a.sp_hist = sp_hist.data(); a.sp_quota = sp_quota.data(); a.sp_node = sp_node.data();
    a.sp_flag = sp_flag.data(); a.sp_oc = sp_oc.data(); a.sp_chunks = CH;
    if ((size_t)kas_fill_lds_layout(sh.n_max, sh.Wc, 1, sh.idmap_entries, sh.need_bsearch, 1).total > lds.size()) return bad("spread fill (LDS)", 0);
    run_fn fa = sh.Wc == 3 ? run_spread_a<3> : sh.Wc == 4 ? run_spread_a<4> : run_spread_a<5>;
    run_fn fb = sh.Wc == 3 ? run_spread_b<3> : sh.Wc == 4 ? run_spread_b<4> : run_spread_b<5>;
    run_fn fp = sh.Wc == 3 ? run_spread_p4<3> : sh.Wc == 4 ? run_spread_p4<4> : run_spread_p4<5>;
    for (int32_t s = 0; s < b->n_scenarios; ++s)
      for (int32_t c = 0; c < CH; ++c) {
        memset(lds.data(), 0xCD, lds.size());
        SpreadArgs ra{&a, s, c, lds.data()};
        if (kasw::run_block(fa, &ra, 1) != 0) return bad("spread fill A", s);
      }
    if (sh.Wc == 3) spread_quota_all<3>(a); else if (sh.Wc == 4) spread_quota_all<4>(a); else spread_quota_all<5>(a);
    for (int32_t s = 0; s < b->n_scenarios; ++s)
      for (int32_t c = 0; c < CH; ++c) {
        memset(lds.data(), 0xCD, lds.size());
        SpreadArgs ra{&a, s, c, lds.data()};
        if (kasw::run_block(fb, &ra, 1) != 0) return bad("spread fill B", s);
      }
    for (int32_t s = 0; s < b->n_scenarios; ++s) {
      memset(lds.data(), 0xCD, lds.size());
      SpreadArgs ra{&a, s, 0, lds.data()};
      if (kasw::run_block(fp, &ra, KAS_SPREAD_P4_WAVES) != 0) return bad("spread fill P4", s);
      g_last_spread += sp_flag[(size_t)s] == 0 ? 1 : 0;
    }
    a.flags |= KAS_FLAG_ONLY_FLAGGED;
  }
  // fill kernel: one workgroup of NW wavefronts per scenario
  run_fn fill = nullptr;
  switch (sh.NW) {
    case 1: fill = fill_for_w<1>(sh.Wc); break;
    case 2: fill = fill_for_w<2>(sh.Wc); break;
    case 8: fill = fill_for_w<8>(sh.Wc); break;
    default: fill = fill_for_w<4>(sh.Wc); break;
  }
  // first fit in a kernel of its own (same decision as kas_solve_device)
  std::vector<int32_t> p4s;
  a.p4s = nullptr;
  // first fit inside the order kernel's workgroup (kas_p4_order_kernel; same decision as kas_launch_plan in kas_hip.hip)
  const bool relax_dual = sh.Wc == 3 && (a.flags & KAS_FLAG_RELAX_DUAL) != 0u;
  const bool relax_idl = !c16 && kas_relax_lds_ids(sh.n_max, sh.any_ctx) && !(getenv("KAS_EMU_RELAX_GATHER") && getenv("KAS_EMU_RELAX_GATHER")[0] == '1');
  // dword mid rows (same decision as kas_plan_mid32 in kas_hip.hip)
  const bool m32 = kas_mid32_launch(sh, c16, flags, relax, a.flags, relax_idl ? 1 : 0, index_rows, CH);
  if (m32) a.flags |= KAS_FLAG_MID32;
  g_last_mid32 = m32 ? 1 : 0;
  // quad tiles (same decision as kas_launch_plan in kas_hip.hip): double tiles + dword mid rows + asked for
  const bool relax_quad = relax_dual && m32 && kas_relax_quad_tiles(flags, b->n_scenarios) && kas_order_relax_lds(sh.n_max, 2, 0, 1) <= KAS_LDS_LIMIT;
  const int relax_tiles = relax_quad ? 2 : (relax_dual ? 1 : 0);
  g_last_relax_quad = relax && relax_quad ? 1 : 0;
  const bool p4_order = relax && kas_p4_with_order(sh, sh.NW, a.flags, CH, b->n_scenarios,
                                                   !sh.any_ctx && (a.flags >> 24) == 0u && (c16 || relax_idl), relax_tiles, relax_idl);
  g_last_p4_order = p4_order ? 1 : 0;
  const bool split_p4 = p4_order || kas_split_p4(sh, sh.NW, a.flags, CH, b->n_scenarios);   // (the fill kernel hands first fit over)
  if (split_p4) {
    p4s.assign((size_t)b->n_topics * (size_t)(KAS_P4S_HEAD + (sh.n_max > 0 ? sh.n_max : 1)) + 64, (int32_t)0xDEADBEEF);
    a.p4s = p4s.data();
    a.flags |= KAS_FLAG_SPLIT_P4;
  } else {
    a.flags &= ~KAS_FLAG_SPLIT_P4;
  }
  g_last_split_p4 = (split_p4 && !p4_order) ? 1 : 0;
  // the slim fill kernel in front (same decision as kas_plan_slim_fill in kas_hip.hip): it writes every scenario's hand-back flag
  const bool slim = KAS_SLIM_FILL_DEFAULT && !(flags & KAS_PLAN_FULL_FILL_BIT) && !c16 && sh.Wc <= 3 && sh.NW == 4 && fused && sh.with_x &&
                    !(flags & KAS_FLAG_NO_RTN_QUOTA) && !index_rows && split_p4 && CH == 0 && sh.idmap_entries > 0 && !sh.need_bsearch;
  g_last_slim_fill = 0;
  if (slim) {
    sp_flag.assign((size_t)b->n_scenarios + 1, (int32_t)0xDEADBEEF);
    a.sp_flag = sp_flag.data();
    run_fn fs = sh.Wc <= 2 ? run_fill_slim<2> : (m32 ? run_fill_slim<3, 1> : run_fill_slim<3>);   // (the product's instance per mid-row layout)
    // exactly the LDS the product launches kas_fill_slim_kernel with (kas_fill_slim_lds), and a guard behind it
    const size_t slim_bytes = (size_t)kas_fill_slim_lds(sh.n_max, sh.Wc, sh.idmap_entries).total;
    std::vector<unsigned char> sl(slim_bytes + 4096);
    for (int32_t s = 0; s < b->n_scenarios; ++s) {
      memset(sl.data(), 0xCD, slim_bytes);
      memset(sl.data() + slim_bytes, 0xA5, 4096);
      RunArgs ra{&a, s, sl.data()};
      if (kasw::run_block(fs, &ra, 4) != 0) return bad("slim fill", s);
      for (size_t i = 0; i < 4096; ++i)
        if (sl[slim_bytes + i] != 0xA5) return bad("slim fill: LDS written beyond kas_fill_slim_lds()", s);
      if (sp_flag[(size_t)s] != 0 && sp_flag[(size_t)s] != 1) return bad("slim fill (hand-back flag not written)", s);
      g_last_slim_fill += sp_flag[(size_t)s] == 0 ? 1 : 0;
    }
    a.flags |= KAS_FLAG_ONLY_FLAGGED;
  }
  g_last_handback = -1;
  {
    const bool by_rank = (a.flags & KAS_FLAG_ONLY_FLAGGED) != 0u && a.sp_flag != nullptr;
    const int32_t failed = emu_launch_fill(fill, a, sh.NW, lds, b->n_scenarios, by_rank);
    if (failed >= 0) return bad("fill (workgroup)", failed);
  }
  for (int32_t s = 0; s < b->n_scenarios; ++s)
    g_last_index_rows += (long)a.stats[(int64_t)s * KAS_STATS_PER_SCENARIO + 6];   // (the fill's own tally, before an order kernel writes there)
  a.flags &= ~KAS_FLAG_ONLY_FLAGGED;
  if (split_p4 && !p4_order) {
    // exactly the LDS the product launches kas_p4_kernel with, and a guard behind it
    const size_t p4_bytes = (size_t)kas_p4_lds_layout(sh.n_max).total;
    std::vector<unsigned char> pl(p4_bytes + 4096);
    run_fn fp4 = sh.Wc <= 2 ? run_p4<2> : sh.Wc == 3 ? (m32 ? run_p4<3, 1> : run_p4<3>) : sh.Wc == 4 ? run_p4<4> : sh.Wc == 5 ? run_p4<5> : run_p4<8>;
    for (int32_t s = 0; s < b->n_scenarios; ++s) {
      memset(pl.data(), 0xCD, p4_bytes);
      memset(pl.data() + p4_bytes, 0xA5, 4096);
      RunArgs ra{&a, s, pl.data()};
      if (kasw::run_block(fp4, &ra, 1) != 0) return bad("first fit (kas_p4_kernel)", s);
      for (size_t i = 0; i < 4096; ++i)
        if (pl[p4_bytes + i] != 0xA5) return bad("first fit: LDS written beyond kas_p4_lds_layout()", s);
    }
  }
  // order kernel: one wavefront per scenario (relaxation form, round form), three per G scenarios (ticket form)
  if (relax && p4_order) {
    run_fn f = m32 ? (relax_quad ? run_p4_order<3, true, false, true, true, true> : relax_dual ? run_p4_order<3, true, false, true, true> : run_p4_order<3, false, false, true, true>)
               : sh.Wc <= 2 ? (c16 ? run_p4_order<2, false, true, false> : run_p4_order<2, false, false, true>)
               : c16 ? (relax_dual ? run_p4_order<3, true, true, false> : run_p4_order<3, false, true, false>)
                     : (relax_dual ? run_p4_order<3, true, false, true> : run_p4_order<3, false, false, true>);
    const size_t fb_bytes = (size_t)kas_p4_order_lds(sh.n_max, relax_tiles, relax_idl);   // exactly the product's LDS, and a guard behind it
    std::vector<unsigned char> rl(fb_bytes + 4096);
    for (int32_t s = 0; s < b->n_scenarios; ++s) {
      memset(rl.data(), 0xCD, fb_bytes);
      memset(rl.data() + fb_bytes, 0xA5, 4096);
      RunArgs ra{&a, s, rl.data()};
      if (kasw::run_block(f, &ra, 2) != 0) return bad("first fit + order (kas_p4_order_kernel)", s);
      for (size_t i = 0; i < 4096; ++i)
        if (rl[fb_bytes + i] != 0xA5) return bad("kas_p4_order_kernel: LDS written beyond kas_p4_order_lds()", s);
      const int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
      g_last_relax_evals += (long)st[9]; g_last_relax_tiles += (long)st[12]; g_last_relax_slow += (long)st[13];
    }
  } else if (relax) {
    const bool rdual = sh.Wc == 3 && (a.flags & KAS_FLAG_RELAX_DUAL) != 0u;
    // the instance kas_order_relax_any (kas_hip.hip) picks: 16-bit cells; int32 cells with the broker ids in the LDS
    // (kas_relax_lds_ids; KAS_EMU_RELAX_GATHER=1 in the environment: the instances that gather them from the node table, which
    // exist without the sampled verification only — as in the product, which refuses the flag there)
    const bool verify = (a.flags >> 24) != 0u;
    const bool idl = !c16 && kas_relax_lds_ids(sh.n_max, sh.any_ctx) && !(getenv("KAS_EMU_RELAX_GATHER") && getenv("KAS_EMU_RELAX_GATHER")[0] == '1');
    if (verify && !c16 && !idl) {
      if (errbuf && errlen > 0) snprintf(errbuf, (size_t)errlen, "KAS_PLAN_VERIFY_SAMPLE: not instantiated for the gather instances");
      return KAS_E_UNSUPPORTED;
    }
    run_fn f = m32 ? (relax_quad ? run_order_relax_m32<true, true> : rdual ? run_order_relax_m32<true> : run_order_relax_m32<false>)
               : c16 ? (verify ? relax_pick<true, true, false>(sh.Wc, rdual, sh.any_ctx) : relax_pick<false, true, false>(sh.Wc, rdual, sh.any_ctx))
               : idl ? (verify ? relax_pick<true, false, true>(sh.Wc, rdual, sh.any_ctx) : relax_pick<false, false, true>(sh.Wc, rdual, sh.any_ctx))
                     : relax_pick<false, false, false>(sh.Wc, rdual, sh.any_ctx);
    g_last_relax_idl = idl ? 1 : 0;
    // exactly the LDS the product launches the kernel with, and a guard behind it: the hardware drops what a
    // workgroup writes beyond its allocation and reads zeros there — here that must not pass unnoticed
    const size_t relax_bytes = (size_t)kas_order_relax_lds(sh.n_max, relax_quad ? 2 : (rdual ? 1 : 0), sh.any_ctx, idl);
    std::vector<unsigned char> rl(relax_bytes + 4096);
    for (int32_t s = 0; s < b->n_scenarios; ++s) {
      memset(rl.data(), 0xCD, relax_bytes);
      memset(rl.data() + relax_bytes, 0xA5, 4096);
      RunArgs ra{&a, s, rl.data()};
      if (kasw::run_block(f, &ra, 1) != 0) return bad("order (relaxation)", s);
      for (size_t i = 0; i < 4096; ++i)
        if (rl[relax_bytes + i] != 0xA5) return bad("order (relaxation): LDS written beyond kas_order_relax_lds()", s);
      const int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
      g_last_relax_evals += (long)st[9]; g_last_relax_tiles += (long)st[12]; g_last_relax_slow += (long)st[13];
    }
  } else if (relaxw) {
    run_fn f = sh.Wc == 4 ? run_order_relax_wide<4> : run_order_relax_wide<5>;
    const size_t rw_bytes = (size_t)kas_order_relaxw_lds(sh.n_max, sh.Wc);      // exactly the product's LDS, and a guard behind it
    std::vector<unsigned char> rl(rw_bytes + 4096);
    for (int32_t s = 0; s < b->n_scenarios; ++s) {
      memset(rl.data(), 0xCD, rw_bytes);
      memset(rl.data() + rw_bytes, 0xA5, 4096);
      RunArgs ra{&a, s, rl.data()};
      if (kasw::run_block(f, &ra, 1) != 0) return bad("order (relaxation, wide lists)", s);
      for (size_t i = 0; i < 4096; ++i)
        if (rl[rw_bytes + i] != 0xA5) return bad("order (relaxation, wide lists): LDS written beyond kas_order_relaxw_lds()", s);
      const int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
      g_last_relax_evals += (long)st[9]; g_last_relax_tiles += (long)st[12];
    }
  } else if (tickets) {
    if (sh.G > 1 && b->n_scenarios > sh.G) {
      a.perm = perm.data();                              // the permutation kernel: one workgroup
      memset(lds.data(), 0xCD, lds.size());
      RunArgs ra{&a, 0, lds.data()};
      if (kasw::run_block(run_permutation, &ra, KAS_PERM_WAVES) != 0) return bad("permutation", 0);
      std::vector<char> seen((size_t)b->n_scenarios, 0);  // it must be a permutation, whatever the order
      for (int32_t j = 0; j < b->n_scenarios; ++j) {
        const int32_t v = perm[(size_t)j];
        if (v < 0 || v >= b->n_scenarios || seen[(size_t)v]) return bad("permutation (not a permutation)", j);
        seen[(size_t)v] = 1;
      }
      int32_t kmax = 0, shift = 0;                       // ... and largest key classes first
      for (int32_t j = 0; j < b->n_scenarios; ++j) kmax = a.scenario_results[j].moved_replicas > kmax ? a.scenario_results[j].moved_replicas : kmax;
      while ((kmax >> shift) >= KAS_PERM_BINS) ++shift;
      for (int32_t j = 0; j + 1 < b->n_scenarios; ++j)
        if ((a.scenario_results[perm[(size_t)j]].moved_replicas >> shift) < (a.scenario_results[perm[(size_t)j + 1]].moved_replicas >> shift))
          return bad("permutation (not descending by key class)", j);
    }
    const bool pk = sh.packed_ok && !(flags & KAS_FLAG_WIDE_COUNTERS);
    run_fn f = sh.G == 1 ? (pk ? tickets_for_g<1, true>(sh.Wc) : tickets_for_g<1, false>(sh.Wc))
             : sh.G == 2 ? (pk ? tickets_for_g<2, true>(sh.Wc) : tickets_for_g<2, false>(sh.Wc))
                         : (pk ? tickets_for_g<4, true>(sh.Wc) : tickets_for_g<4, false>(sh.Wc));
    for (int32_t s = 0; s < b->n_scenarios; s += sh.G) {
      memset(lds.data(), 0xCD, lds.size());
      RunArgs ra{&a, s, lds.data()};
      if (kasw::run_block(f, &ra, 3) != 0) return bad("order (tickets)", s);
    }
    for (int32_t s = 0; s < b->n_scenarios; ++s) g_last_queue_rows += (long)a.stats[(int64_t)s * KAS_STATS_PER_SCENARIO + 14];
    if (getenv("KAS_EMU_STATS")) {
      for (int32_t s = 0; s < b->n_scenarios; ++s) {
        const int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
        fprintf(stderr, "emu stats s=%d solver_iter=%lld queue_passes=%lld run_rounds=%lld run_rows=%lld blocked=%lld stager_iter=%lld\n", s,
                (long long)st[9], (long long)st[6], (long long)st[10], (long long)st[14], (long long)st[11], (long long)st[12]);
      }
    }
  } else if (wide) {
    if (sh.wide_checked) a.flags |= KAS_FLAG_WIDE_CHECK;   // as kas_solve_device
    run_fn f = sh.Wc == 4 ? run_order_wide<4> : run_order_wide<5>;
    for (int32_t s = 0; s < b->n_scenarios; ++s) {
      memset(lds.data(), 0xCD, lds.size());
      RunArgs ra{&a, s, lds.data()};
      if (kasw::run_block(f, &ra, KAS_ORDER_WIDE_BLOCK / 64) != 0) return bad("order (wide tickets)", s);
    }
    for (int32_t s = 0; s < b->n_scenarios; ++s) g_last_queue_rows += (long)a.stats[(int64_t)s * KAS_STATS_PER_SCENARIO + 14];
    if (getenv("KAS_EMU_STATS")) {
      for (int32_t s = 0; s < b->n_scenarios; ++s) {
        const int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
        fprintf(stderr, "emu stats (wide) s=%d solver_iter=%lld bulk_solver_iter=%lld queue_passes=%lld run_rounds=%lld run_rows=%lld blocked=%lld stager_iter=%lld stager_idle=%lld sched_rounds(last block)=%ld\n", s,
                (long long)st[9], (long long)st[15], (long long)st[6], (long long)st[10], (long long)st[14], (long long)st[11], (long long)st[12],
                (long long)st[13], kasw::g_last_block_rounds);
        fprintf(stderr, "emu diag (wide, -DKAS_WIDE_DIAG) joint_steps=%lld in_hand=%lld hold_hot=%lld wait_hot_only=%lld eligible=%lld | not eligible: many_ahead_elsewhere=%lld one_ahead_not_in_hand=%lld behind_gap=%lld\n",
                (long long)st[13], (long long)st[3], (long long)st[4], (long long)st[5], (long long)st[7], (long long)st[0], (long long)st[1], (long long)st[2]);
      }
    }
  } else {
    run_fn f = rounds_for(sh.Wc);
    for (int32_t s = 0; s < b->n_scenarios; ++s) {
      memset(lds.data(), 0xCD, lds.size());
      RunArgs ra{&a, s, lds.data()};
      if (kasw::run_block(f, &ra, 1) != 0) return bad("order (rounds)", s);
    }
  }
  const bool wide_recheck = wide && sh.wide_checked;
  if (wide_recheck) {
    // as kas_solve_device: a scenario whose counts outgrew the wide form's fields is filled again ...
    KasLaunch af = a;
    af.flags = (af.flags | KAS_FLAG_ONLY_FLAGGED) & ~(KAS_FLAG_WIDE_CHECK | KAS_FLAG_SPLIT_P4);
    af.sp_flag = ord_flag.data();
    {
      const int32_t failed = emu_launch_fill(fill, af, sh.NW, lds, b->n_scenarios, false);
      if (failed >= 0) return bad("fill (flagged by the wide form; workgroup)", failed);
    }
    a.flags &= ~KAS_FLAG_WIDE_CHECK;
  }
  if ((sh.any_ctx && (tickets || wide || relax)) || wide_recheck) {
    // as kas_solve_device: the round form behind a ticket form, taking only what that one flagged
    a.flags |= KAS_FLAG_ORDER_FLAGGED;
    a.perm = nullptr;
    run_fn f = rounds_for(sh.Wc);
    for (int32_t s = 0; s < b->n_scenarios; ++s) {
      g_last_flagged += ord_flag[(size_t)s] != 0 ? 1 : 0;
      memset(lds.data(), 0xCD, lds.size());
      RunArgs ra{&a, s, lds.data()};
      if (kasw::run_block(f, &ra, 1) != 0) return bad("order (rounds, flagged)", s);
    }
  }
  return KAS_E_OK;
}

// The product's planning decision for a batch shape, without running anything (plan-math tests):
// out[0..9] = tickets_ok, wide_ok, round_fits, G, NW, with_x, packed_ok, fused_ok, wide_checked, relax_ok.  Returns kas_shape_batch's code.
extern "C" __attribute__((visibility("default")))
int kas_emu_shape(const kas_batch_desc* b, int32_t* out, char* errbuf, int errlen) {
  KasShape sh;
  std::string err;
  const int rc = kas_shape_batch(b, &sh, &err, 0, 0);
  if (rc != KAS_E_OK) {
    if (errbuf && errlen > 0) { strncpy(errbuf, err.c_str(), (size_t)errlen - 1); errbuf[errlen - 1] = 0; }
    return rc;
  }
  out[0] = sh.tickets_ok; out[1] = sh.wide_ok; out[2] = sh.round_fits; out[3] = sh.G; out[4] = sh.NW;
  out[5] = sh.with_x; out[6] = sh.packed_ok; out[7] = sh.fused_ok; out[8] = sh.wide_checked; out[9] = sh.relax_ok;
  return rc;
}

// spread fill planning: chunks per scenario for a batch of n_scenarios single-topic scenarios of this shape
// (0: the one-workgroup fill kernel), and the LDS of its scan kernels (pass A, pass B, the one-workgroup layout)
extern "C" __attribute__((visibility("default")))
int kas_emu_spread_plan(const kas_batch_desc* b, int32_t* out) {
  KasShape sh;
  std::string err;
  const int rc = kas_shape_batch(b, &sh, &err, 0, 0);
  if (rc != KAS_E_OK) return rc;
  out[0] = kas_spread_chunks(sh, b->n_scenarios, kas_batch_single_topic(b), false);
  out[1] = kas_spread_scan_lds(sh.n_max, sh.Wc, sh.idmap_entries, sh.need_bsearch, 1).total;
  out[2] = kas_spread_scan_lds(sh.n_max, sh.Wc, sh.idmap_entries, sh.need_bsearch, 2).total;
  out[3] = kas_fill_lds_layout(sh.n_max, sh.Wc, 1, sh.idmap_entries, sh.need_bsearch, 1).total;
  out[4] = (int32_t)(((int64_t)sh.max_partitions + 63) / 64);
  return rc;
}

extern "C" __attribute__((visibility("default")))
long kas_emu_collectives(void) { return kasw::g_emu.collectives; }

extern "C" __attribute__((visibility("default")))
long kas_emu_last_queue_rows(void) { return g_last_queue_rows; }

extern "C" __attribute__((visibility("default")))
int kas_emu_last_fused(void) { return g_last_fused; }

extern "C" __attribute__((visibility("default")))
long kas_emu_last_index_rows(void) { return g_last_index_rows; }
extern "C" __attribute__((visibility("default")))
long kas_emu_last_mid32(void) { return g_last_mid32; }
extern "C" __attribute__((visibility("default")))
long kas_emu_last_relax_quad(void) { return g_last_relax_quad; }
extern "C" __attribute__((visibility("default")))
long kas_emu_last_slim_fill(void) { return g_last_slim_fill; }

// scenarios the last by-rank launch of kas_fill_kernel counted as flagged (KasLaunch::handback; -1: no such launch in the last solve)
extern "C" __attribute__((visibility("default")))
int kas_emu_last_handback(void) { return g_last_handback; }

extern "C" __attribute__((visibility("default")))
int kas_emu_last_relax_idl(void) { return g_last_relax_idl; }

extern "C" __attribute__((visibility("default")))
int kas_emu_last_p4_order(void) { return g_last_p4_order; }

extern "C" __attribute__((visibility("default")))
int kas_emu_last_spread(void) { return g_last_spread; }

extern "C" __attribute__((visibility("default")))
int kas_emu_last_flagged(void) { return g_last_flagged; }

extern "C" __attribute__((visibility("default")))
int kas_emu_last_order_form(void) { return g_last_order_form; }
// 1: the last solve's first fit ran in kas_p4_kernel
extern "C" __attribute__((visibility("default")))
int kas_emu_last_split_p4(void) { return g_last_split_p4; }


// relaxation form of the last kas_emu_solve_batch: out[0..2] = tiles, evaluations, tiles off the straight-line path
extern "C" __attribute__((visibility("default")))
void kas_emu_last_relax_stats(long* out) { out[0] = g_last_relax_tiles; out[1] = g_last_relax_evals; out[2] = g_last_relax_slow; }

// ---------------------------------------------------------------------------------------------
// Unit harness for the parallel P4 of the fill kernel (p4_lists_parallel<3, 4>): the caller gives the
// node state (load, rack, the list of non-full nodes in processing order), the orphan rows in row
// order and their mid rows; the four wavefronts run the windows exactly as the kernel does.  With
// KAS_EMU_WAVE_DIV one wave can be made slow, which is how a test provokes one window overtaking
// another (tests/test_emu_p4_windows.py).  Returns 0, or -100 on divergence / deadlock.
// ---------------------------------------------------------------------------------------------
namespace {
struct P4Args { kas::LdsView L; kas::TopicView T; int32_t live_count; int32_t fail_row; };
void run_p4_unit(void* p) {
  P4Args* r = (P4Args*)p;
  int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int32_t fail_win = -1, fail_row = -1;
  kas::p4_lists_parallel<3, 4>(r->L, r->T, r->live_count, kasw::wave_id(), st, fail_win, fail_row);
  if (fail_win >= 0 && kasw::lane() == 0) r->fail_row = fail_row;     // (any failing window: the tests place everything)
}
}  // namespace

extern "C" __attribute__((visibility("default")))
int kas_emu_p4_unit(int32_t n_nodes, const int32_t* load, const int32_t* rack, int32_t cap, int32_t live_count,
                    const int32_t* live, int32_t n_orphans, int32_t* orphan_rows, int32_t P, uint16_t* mid,
                    int32_t* load_out) {
  const KasLds lay = kas_fill_lds_layout(n_nodes, 3, 4, 0, 0, 1);
  std::vector<unsigned char> lds((size_t)lay.total + 64, 0xCD);
  P4Args r;
  r.L.x = (int32_t*)(lds.data() + lay.off_x);
  r.L.load = (int32_t*)(lds.data() + lay.off_load);
  r.L.qrs = (int32_t*)(lds.data() + lay.off_qrs);
  r.L.rack = (int16_t*)(lds.data() + lay.off_rack);
  r.L.live = (int16_t*)(lds.data() + lay.off_live);
  r.L.ns = 1; r.L.rs = 1;
  r.L.idmap = (int16_t*)(lds.data() + lay.off_idmap);
  r.L.ids = (int32_t*)(lds.data() + lay.off_ids);
  r.L.ring_p = (int32_t*)(lds.data() + lay.off_ring);
  r.L.ring_meta = r.L.ring_p + KAS_RING_CAP;
  r.L.ring_rack = (int16_t*)(r.L.ring_meta + KAS_RING_CAP);
  r.L.ctl = (int32_t*)(lds.data() + lay.off_ctl);
  for (int32_t i = 0; i < n_nodes; ++i) { r.L.load[i] = load[i]; r.L.rack[i] = (int16_t)rack[i]; }
  for (int32_t i = 0; i < live_count; ++i) r.L.live[i] = (int16_t)live[i];
  for (int i = 0; i < KAS_CTL_INTS; ++i) r.L.ctl[i] = i == KAS_CTL_FAILROW ? -1 : (i == KAS_CTL_FAILWIN ? 0x7fffffff : 0);
  r.L.ctl[KAS_CTL_OC] = n_orphans;                               // one list: chunk 0 holds every orphan
  r.L.ctl[KAS_CTL_LIVE] = live_count;
  memset(&r.T, 0, sizeof(r.T));
  r.T.orph = orphan_rows; r.T.mid = mid;
  r.T.P = P; r.T.cw = 3; r.T.rf = 3; r.T.ow = 3; r.T.nt = (P + 63) >> 6; r.T.N = n_nodes; r.T.cap = cap;
  r.live_count = live_count; r.fail_row = -1;
  if (kasw::run_block(run_p4_unit, &r, 4) != 0) return -100;
  for (int32_t i = 0; i < n_nodes; ++i) load_out[i] = r.L.load[i];
  return r.fail_row >= 0 ? 1 : 0;
}
