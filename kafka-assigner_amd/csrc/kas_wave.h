// kas_wave.h — wavefront / workgroup primitives used by the solver body, gfx950 implementation.
//
// One workgroup == one scenario == NW 64-lane wavefronts sharing the scenario's LDS state.
// Cross-lane steps are wave-level operations (64-bit ballots, lane shuffles, LDS atomics);
// kasw::sync() is the workgroup barrier between phases (it carries the workgroup-scope
// release/acquire that orders LDS and same-CU global accesses: out rows, orphan lists and
// accept-mask words written by one wave are re-read by another wave of the same workgroup,
// which shares the CU's vector L1).  The three waves of the order kernel talk through LDS slot
// tags only and poll them with repoll().
//
// tests/emu/kas_wave.h provides the same names on top of CPU fibers so the identical body
// source can be stepped on a machine without a GPU; the product only ever includes this file.
#ifndef KAS_WAVE_H_
#define KAS_WAVE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KAS_DEV __device__ __forceinline__
// rarely taken paths: a real call, so that their registers do not count against the hot loops
#define KAS_DEV_COLD __device__ __noinline__

namespace kasw {

KAS_DEV int lane() { return (int)(threadIdx.x & 63u); }

KAS_DEV int tid() { return (int)threadIdx.x; }

// wavefront index inside the workgroup (wave-uniform, kept in an SGPR)
KAS_DEV int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// (the builtin takes the predicate itself: __ballot(int) made the compiler materialise 0 / 1 in a VGPR
// and compare it again wherever the predicate was an AND of lane masks)
KAS_DEV uint64_t ballot(bool p) { return (uint64_t)__builtin_amdgcn_ballot_w64(p); }

KAS_DEV int shfl(int v, int src_lane) { return __shfl(v, src_lane, 64); }

// value of v in one lane, the same lane for the whole wavefront (v_readlane: no LDS crossbar trip)
KAS_DEV int read_lane(int v, int uniform_lane) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(uniform_lane));
}

// a value every lane of the wavefront holds alike, moved to a scalar register (branches on it are scalar branches)
KAS_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// a pointer every lane holds alike, moved to scalar registers (the base of a global_load with a per-lane offset)
template <class T>
KAS_DEV T* uniform_ptr(T* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (T*)(((uint64_t)hi << 32) | lo);
}

KAS_DEV void sync() { __syncthreads(); }

// Ordering point for a section that only ONE wave of the workgroup executes: this wave's
// earlier LDS and global accesses complete before its later ones are issued (no s_barrier, the
// other waves are not coming).
KAS_DEV void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// back off inside a spin loop (the other waves of the CU get the issue slots): ~64 cycles, or
// ~64 * N for a wave that is not on anybody's critical path
KAS_DEV void spin_pause() { __builtin_amdgcn_s_sleep(1); }
template <int N>
KAS_DEV void nap() { __builtin_amdgcn_s_sleep(N); }

// issue priority of this wave among the waves of its SIMD (0..3): the one wave whose loop is a
// dependency chain asks for the slots first
template <int P>
KAS_DEV void set_priority() { __builtin_amdgcn_s_setprio(P); }

// Point where the body relies on the 64 lanes having executed the preceding LDS accesses before
// any lane executes the following ones (read-then-overwrite, atomic-then-read on the same words).
// A wavefront issues each instruction for all lanes at once and the LDS serves a wave's
// operations in issue order, so on hardware this only has to stop the COMPILER from reordering
// or caching memory operations across it: no instruction, no s_waitcnt.
KAS_DEV void lockstep() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Top of a spin loop that polls LDS written by ANOTHER wave of the workgroup: the compiler must
// re-read LDS after this point, the hardware needs nothing (LDS has no cache in front of it and
// serves the operations of a wave in issue order).
KAS_DEV void repoll() { asm volatile("" ::: "memory"); }

// Identity the optimiser cannot see through.  Used on the elements of small register tables
// before a select chain: without it LLVM folds "select(i==j, a[j], ...)" back into a
// dynamically indexed load and the table moves to scratch (= global memory).
KAS_DEV int32_t opaque(int32_t v) {
  asm("" : "+v"(v));
  return v;
}

// v_perm_b32: result byte i = byte sel[i] of the eight bytes hi:lo (0..3 = lo, 4..7 = hi), 0x0c = 0x00
KAS_DEV uint32_t perm_bytes(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// A value that must be in its register at this point of the program: a load feeding it is issued before, and cannot be
// sunk into the branch that uses it (the compiler does that to a load whose result only one side of a test needs —
// which puts the load's latency behind the test's).
KAS_DEV int32_t pinned(int32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}

// a * b for small non-negative factors (node index x block stride): the full-rate 24-bit multiply
KAS_DEV int32_t mul24(int32_t a, int32_t b) { return (int32_t)__umul24((unsigned)a, (unsigned)b); }

KAS_DEV int popc(uint64_t m) { return __popcll((unsigned long long)m); }

// index of the lowest set bit; m must be non-zero
KAS_DEV int first_lane(uint64_t m) { return __ffsll((unsigned long long)m) - 1; }

KAS_DEV uint64_t lanemask_lt() { return (1ull << lane()) - 1ull; }

// popc(m & lanemask_lt()): bits of m below this lane (v_mbcnt_lo / v_mbcnt_hi: two instructions, m may
// differ per lane)
KAS_DEV int count_below(uint64_t m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

KAS_DEV int lds_atomic_add(int* p, int v) { return atomicAdd(p, v); }
KAS_DEV void lds_atomic_min(int* p, int v) { atomicMin(p, v); }
KAS_DEV void lds_atomic_max(int* p, int v) { atomicMax(p, v); }

// counters in HBM that several workgroups add to (spread fill: movement counts of a scenario's chunks)
KAS_DEV void global_atomic_add(int* p, int v) { atomicAdd(p, v); }

KAS_DEV void lds_atomic_or_u32(uint32_t* p, uint32_t v) {
  __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

KAS_DEV void lds_atomic_or_u64(uint64_t* p, uint64_t v) {
  __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

KAS_DEV void lds_atomic_add_u64(uint64_t* p, uint64_t v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// LDS add with return, add / subtract without (ds_add_rtn_u32, ds_add_u32, ds_sub_u32).  The relaxation form of
// the order kernel (kas_order_relax.h) uses the returned value as a PREFIX SUM: when several lanes of one
// wavefront instruction name the same word, lane i gets the word's value before the instruction plus the
// addends of the lanes BELOW i naming it.  That the LDS serves the lanes of one instruction in ascending lane
// order is not in the ISA documents; it is measured (tools/lds_order_probe.hip, tools/issue_probe.hip: 0 of
// 2 x 10^10 lane-operations out of order, 32- and 64-bit, busy LDS or not) and checked again by
// kas_ctx_create's self-test, which keeps the form off a device that does not pass.
KAS_DEV uint32_t lds_add_rtn_u32(uint32_t* p, uint32_t v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
KAS_DEV void lds_add_u32(uint32_t* p, uint32_t v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
KAS_DEV void lds_sub_u32(uint32_t* p, uint32_t v) {
  (void)__hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// ... and on 64-bit words (ds_add_rtn_u64, ds_sub_u64): the relaxation form for lists 4 and 5 wide (kas_order_relax_wide.h), same
// property (the probes measured both widths)
KAS_DEV uint64_t lds_add_rtn_u64(uint64_t* p, uint64_t v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
KAS_DEV void lds_sub_u64(uint64_t* p, uint64_t v) {
  (void)__hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Global loads the COMPILER DOES NOT SEE (inline assembly): gload_*_async asks for a value, wait_loads() is the one
// s_waitcnt vmcnt(0) that makes everything asked for so far usable, arrived(x) ties a use of x behind it.  Why not plain
// loads: the compiler's wait insertion is exact inside a basic block only.  A loaded value that lives across a loop's back
// edge in a loop with several paths gets "vmcnt(0)" at the latch (a register copy of the value is a use), i.e. behind the
// requests the iteration has just made — the order kernel's one wavefront then waits out an L2 or HBM round trip per tile
// on its dependency chain.  With these the kernel decides where it waits: once per step, where every outstanding request
// is a whole step old (kas_order_relax.h).  The value lands in `dst` itself ("+v": the register of the loop-carried
// variable), so there is no copy to wait for.  base: wave-uniform (scalar registers), voff: byte offset per lane.
template <int IMM = 0>
KAS_DEV void gload_u32_async(uint32_t& dst, const void* base, uint32_t voff) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
template <int IMM = 0>
KAS_DEV void gload_u16_async(uint32_t& dst, const void* base, uint32_t voff) {
  asm volatile("global_load_ushort %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}
// ... for the lanes with `on` only: the instruction is issued under a narrowed EXEC (restored behind it), so a lane that does
// not ask keeps what its register holds and nothing is requested for it — the statement itself is unconditional, which is
// what keeps the compiler from merging two definitions of `dst` with a copy (kas_order_relax.h, mid_request)
template <int IMM = 0>
KAS_DEV void gload_u32_async_if(uint32_t& dst, const void* base, uint32_t voff, bool on) {
  const uint64_t m = (uint64_t)__builtin_amdgcn_ballot_w64(on);
  uint64_t saved;
  asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %2\n\tglobal_load_dword %0, %3, %4 offset:%5\n\ts_mov_b64 exec, %1"
               : "+v"(dst), "=&s"(saved) : "s"(m), "v"(voff), "s"(base), "n"(IMM) : "memory");
}
KAS_DEV void wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
KAS_DEV void arrived(uint32_t& x) { asm volatile("" : "+v"(x)); }

// 64-bit word written earlier by this wave (accept-mask scratch): force a vector load so the
// value never comes from the scalar cache, which is not coherent with the wave's own stores.
KAS_DEV uint64_t load_shared_u64(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
KAS_DEV void store_shared_u64(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// ... and an 8-byte word of LDS one wavefront of the workgroup writes and another polls (kas_p4_order_kernel: first fit's progress
// for the order wavefront): one ds_write_b64 / ds_read_b64, so that the two halves are never seen apart
KAS_DEV uint64_t load_shared_u64_lds(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
KAS_DEV void store_shared_u64_lds(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// A word of LDS another wave of the workgroup writes / reads (list counts, flags): a relaxed atomic access, NOT a volatile
// one — `*(volatile uint32_t*)p` on a pointer the compiler holds as generic is a FLAT load (sc0 sc1) that the wave then
// waits for with vmcnt(0); the atomic form keeps the LDS address space and becomes ds_read_b32 / ds_write_b32.  Re-read
// on every call: callers put kasw::repoll() where the order against other LDS accesses matters.
KAS_DEV uint32_t load_shared_u32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
KAS_DEV void store_shared_u32(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// constant 100 MHz device clock (s_memrealtime): 10 ns ticks
KAS_DEV int64_t clock_ticks() { return (int64_t)wall_clock64(); }

KAS_DEV int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

KAS_DEV uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64);
    uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
    v += ((uint64_t)hi << 32) | lo;
  }
  return v;
}

}  // namespace kasw
#endif  // KAS_WAVE_H_
