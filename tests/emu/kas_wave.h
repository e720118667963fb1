// tests/emu/kas_wave.h — CPU stand-in for kafka-assigner_amd/csrc/kas_wave.h.  TEST INFRASTRUCTURE.
//
// Lets the unmodified solver body (csrc/kas_solver_body.h) be compiled with g++ and stepped on
// a machine without a GPU: the 64 * NW lanes of a workgroup are ucontext fibers run by a single
// thread.  Every wave collective (ballot, shfl, lockstep, wave_sync, reductions) is a rendezvous
// of the 64 fibers of that wave: a fiber parks there until all 64 have arrived, so lane i always
// observes what lanes < i did before the rendezvous and nothing they do after it — the lockstep
// a real wavefront provides.  kasw::sync() is a rendezvous of all fibers of the workgroup.
// Waves are otherwise free-running relative to each other (round-robin), so cross-wave spin
// loops (ticket waits, the watermark) make progress as long as they contain a wave collective.
// The scheduler aborts when lanes of a wave arrive at different kinds of collective (= divergent
// control flow around a collective, which would hang or corrupt on hardware) or when nothing can
// run (deadlock).
//
// It checks the LOGIC of the kernel source against the oracle before GPU minutes are spent.
// It is not a product path: nothing under kafka-assigner_amd/ includes or links it.
#ifndef KAS_WAVE_H_
#define KAS_WAVE_H_
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <ucontext.h>

#define KAS_DEV static inline
#define KAS_DEV_COLD static
#define KAS_EMU_MAX_LANES 1024

namespace kasw {

enum Kind { K_NONE = 0, K_BALLOT, K_SHFL, K_SYNC, K_LOCKSTEP, K_SUM, K_SUM64, K_WAVESYNC };
enum State { S_RUNNABLE = 0, S_PARKED, S_DONE };

// Fiber switch.  x86-64: six callee-saved registers and the stack pointer, no system call
// (glibc's swapcontext saves the signal mask with one per switch: a third of an emulated solve).
// Elsewhere, or with -DKAS_EMU_UCONTEXT: ucontext.
#if defined(__x86_64__) && !defined(KAS_EMU_UCONTEXT)
#define KAS_EMU_FAST_SWITCH 1
struct Ctx { void* sp; };
extern "C" void kas_emu_switch(Ctx* from, Ctx* to);
#else
#define KAS_EMU_FAST_SWITCH 0
struct Ctx { ucontext_t uc; };
KAS_DEV void kas_emu_switch(Ctx* from, Ctx* to) { swapcontext(&from->uc, &to->uc); }
#endif

struct Emu {
  Ctx main_ctx;
  Ctx lane_ctx[KAS_EMU_MAX_LANES];
  int state[KAS_EMU_MAX_LANES];
  int kind[KAS_EMU_MAX_LANES];
  int n_lanes;
  int cur;               // fiber being run
  uint64_t slot[KAS_EMU_MAX_LANES];
  uint64_t slot2[KAS_EMU_MAX_LANES];
  long collectives;
};

extern Emu g_emu;

KAS_DEV int lane() { return g_emu.cur & 63; }
KAS_DEV int tid() { return g_emu.cur; }
KAS_DEV int wave_id() { return g_emu.cur >> 6; }

// park this fiber at a collective of kind k; returns when its group has arrived
KAS_DEV void rendezvous(int k) {
  Emu& e = g_emu;
  e.kind[e.cur] = k;
  e.state[e.cur] = S_PARKED;
  kas_emu_switch(&e.lane_ctx[e.cur], &e.main_ctx);
}

KAS_DEV uint64_t ballot(bool p) {
  const int base = g_emu.cur & ~63;
  g_emu.slot[g_emu.cur] = p ? 1 : 0;
  rendezvous(K_BALLOT);
  uint64_t m = 0;
  for (int i = 0; i < 64; ++i) m |= (g_emu.slot[base + i] & 1ull) << i;
  rendezvous(K_BALLOT);   // everyone has read the slots before they are reused
  return m;
}

KAS_DEV int shfl(int v, int src_lane) {
  const int base = g_emu.cur & ~63;
  g_emu.slot[g_emu.cur] = (uint64_t)(uint32_t)v;
  rendezvous(K_SHFL);
  int r = (int)(uint32_t)g_emu.slot[base + (src_lane & 63)];
  rendezvous(K_SHFL);
  return r;
}

KAS_DEV int read_lane(int v, int uniform_lane) { return shfl(v, uniform_lane); }

KAS_DEV int uniform(int v) { return v; }                  // (hardware: v_readfirstlane)

template <class T>
KAS_DEV T* uniform_ptr(T* p) { return p; }                  // (hardware: both halves through v_readfirstlane)
KAS_DEV void sync() { rendezvous(K_SYNC); }
KAS_DEV void lockstep() { rendezvous(K_LOCKSTEP); }
KAS_DEV void wave_sync() { rendezvous(K_WAVESYNC); }
KAS_DEV void spin_pause() {}
template <int N> KAS_DEV void nap() {}
template <int P> KAS_DEV void set_priority() {}
KAS_DEV void repoll() { rendezvous(K_LOCKSTEP); }   // lets the other waves run

KAS_DEV int32_t opaque(int32_t v) { return v; }
KAS_DEV int32_t pinned(int32_t v) { return v; }               // (hardware: a value the compiler must have in a register HERE)

KAS_DEV int32_t mul24(int32_t a, int32_t b) { return a * b; }
// (hardware: v_perm_b32; selectors 0..7 and 0x0c are the ones the kernels use)
KAS_DEV uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel) {
  const uint64_t src = ((uint64_t)hi << 32) | lo;
  uint32_t r = 0;
  for (int i = 0; i < 4; ++i) {
    const uint32_t s = (sel >> (8 * i)) & 0xffu;
    const uint32_t b = s < 8u ? (uint32_t)((src >> (8 * s)) & 0xffu) : (s == 0x0cu ? 0u : 0xffu);
    r |= b << (8 * i);
  }
  return r;
}
KAS_DEV int popc(uint64_t m) { return __builtin_popcountll(m); }
KAS_DEV int first_lane(uint64_t m) { return __builtin_ctzll(m); }
KAS_DEV uint64_t lanemask_lt() { return (1ull << lane()) - 1ull; }
KAS_DEV int count_below(uint64_t m) { return __builtin_popcountll(m & lanemask_lt()); }

KAS_DEV int lds_atomic_add(int* p, int v) { int o = *p; *p = o + v; return o; }
KAS_DEV void lds_atomic_min(int* p, int v) { if (v < *p) *p = v; }
KAS_DEV void lds_atomic_max(int* p, int v) { if (v > *p) *p = v; }
KAS_DEV void lds_atomic_or_u32(uint32_t* p, uint32_t v) { *p |= v; }
KAS_DEV void lds_atomic_or_u64(uint64_t* p, uint64_t v) { *p |= v; }
KAS_DEV void lds_atomic_add_u64(uint64_t* p, uint64_t v) { *p += v; }

// (hardware: the lanes of one LDS atomic instruction are served in ascending lane order; here the fibers of a
// wave run one after the other, in lane order, between two rendezvous — the callers put a lockstep() around
// every group of these that must look like one instruction)
#ifdef KAS_EMU_RTN_DESCENDING
// Test build: an LDS that serves the lanes of one atomic-with-return instruction in DESCENDING lane order — the hardware
// the relaxation form of P5 would NOT survive (tests/test_emu_lane_order.py: its lists come out wrong with status OK, and
// KAS_PLAN_VERIFY_SAMPLE turns them into KAS_FAIL_WATCHDOG).  Every lane of the wave must execute the instruction.
KAS_DEV uint32_t lds_add_rtn_u32(uint32_t* p, uint32_t v) {
  Emu& e = g_emu;
  const int base = e.cur & ~63, me = e.cur & 63;
  e.slot[e.cur] = (uint64_t)(uintptr_t)p; e.slot2[e.cur] = v;
  rendezvous(K_LOCKSTEP);
  uint32_t ret = *p;                                         // the word as the instruction found it + the HIGHER lanes' addends
  for (int l = me + 1; l < 64; ++l) if (e.slot[base + l] == (uint64_t)(uintptr_t)p) ret += (uint32_t)e.slot2[base + l];
  rendezvous(K_LOCKSTEP);
  *p += v;
  return ret;
}
#else
KAS_DEV uint32_t lds_add_rtn_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
#endif
KAS_DEV void lds_add_u32(uint32_t* p, uint32_t v) { *p += v; }
KAS_DEV void lds_sub_u32(uint32_t* p, uint32_t v) { *p -= v; }
#ifdef KAS_EMU_RTN_DESCENDING
KAS_DEV uint64_t lds_add_rtn_u64(uint64_t* p, uint64_t v) {   // (as lds_add_rtn_u32 of this build: the lanes served in DESCENDING order)
  Emu& e = g_emu;
  const int base = e.cur & ~63, me = e.cur & 63;
  e.slot[e.cur] = (uint64_t)(uintptr_t)p; e.slot2[e.cur] = v;
  rendezvous(K_LOCKSTEP);
  uint64_t ret = *p;
  for (int l = me + 1; l < 64; ++l) if (e.slot[base + l] == (uint64_t)(uintptr_t)p) ret += e.slot2[base + l];
  rendezvous(K_LOCKSTEP);
  *p += v;
  return ret;
}
#else
KAS_DEV uint64_t lds_add_rtn_u64(uint64_t* p, uint64_t v) { uint64_t o = *p; *p = o + v; return o; }
#endif
KAS_DEV void lds_sub_u64(uint64_t* p, uint64_t v) { *p -= v; }

// (hardware: global loads issued as inline assembly and waited for once per step, csrc/kas_wave.h; here they are loads)
template <int IMM = 0>
KAS_DEV void gload_u32_async(uint32_t& dst, const void* base, uint32_t voff) {
  dst = *(const uint32_t*)((const char*)base + voff + IMM);
}
template <int IMM = 0>
KAS_DEV void gload_u16_async(uint32_t& dst, const void* base, uint32_t voff) {
  dst = (uint32_t)*(const uint16_t*)((const char*)base + voff + IMM);
}
template <int IMM = 0>
KAS_DEV void gload_u32_async_if(uint32_t& dst, const void* base, uint32_t voff, bool on) {
  if (on) dst = *(const uint32_t*)((const char*)base + voff + IMM);
}
KAS_DEV void wait_loads() {}
KAS_DEV void arrived(uint32_t&) {}

KAS_DEV uint32_t load_shared_u32(const uint32_t* p) { return *(const volatile uint32_t*)p; }
KAS_DEV void store_shared_u32(uint32_t* p, uint32_t v) { *(volatile uint32_t*)p = v; }
KAS_DEV uint64_t load_shared_u64(const uint64_t* p) { return *(const volatile uint64_t*)p; }
KAS_DEV void store_shared_u64(uint64_t* p, uint64_t v) { *(volatile uint64_t*)p = v; }
KAS_DEV uint64_t load_shared_u64_lds(const uint64_t* p) { return *(const volatile uint64_t*)p; }
KAS_DEV void store_shared_u64_lds(uint64_t* p, uint64_t v) { *(volatile uint64_t*)p = v; }

KAS_DEV int64_t clock_ticks() { return 0; }

KAS_DEV void global_atomic_add(int* p, int v) { *p += v; }   // (one block runs at a time)

KAS_DEV int wave_sum(int v) {
  const int base = g_emu.cur & ~63;
  g_emu.slot[g_emu.cur] = (uint64_t)(int64_t)v;
  rendezvous(K_SUM);
  int64_t s = 0;
  for (int i = 0; i < 64; ++i) s += (int64_t)g_emu.slot[base + i];
  rendezvous(K_SUM);
  return (int)s;
}

KAS_DEV uint64_t wave_sum_u64(uint64_t v) {
  const int base = g_emu.cur & ~63;
  g_emu.slot[g_emu.cur] = v;
  rendezvous(K_SUM64);
  uint64_t s = 0;
  for (int i = 0; i < 64; ++i) s += g_emu.slot[base + i];
  rendezvous(K_SUM64);
  return s;
}

// Run fn(arg) as n_waves * 64 fibers.  Returns 0, or -1 on divergence / deadlock.
int run_block(void (*fn)(void*), void* arg, int n_waves);

}  // namespace kasw
#endif  // KAS_WAVE_H_
