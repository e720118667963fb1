// kas_wave.h — wavefront primitives used by the solver body, gfx950 implementation.
//
// One workgroup == one 64-lane wavefront == one scenario, so every cross-lane step is a
// wave-level operation: 64-bit ballots, lane shuffles and LDS atomics.  kasw::sync() is the
// only ordering primitive the body uses; with a single-wave workgroup s_barrier is nearly free
// and __syncthreads() carries the workgroup-scope release/acquire that orders both LDS and
// same-CU global accesses (the accept-mask scratch and the out rows are re-read by the wave
// that wrote them).
//
// tests/emu/kas_wave.h provides the same names on top of CPU fibers so the identical body
// source can be stepped on a machine without a GPU; the product only ever includes this file.
#ifndef KAS_WAVE_H_
#define KAS_WAVE_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KAS_DEV __device__ __forceinline__

namespace kasw {

KAS_DEV int lane() { return (int)(threadIdx.x & 63u); }

KAS_DEV uint64_t ballot(bool p) { return (uint64_t)__ballot(p ? 1 : 0); }

KAS_DEV int shfl(int v, int src_lane) { return __shfl(v, src_lane, 64); }

KAS_DEV void sync() { __syncthreads(); }

// Point where the body relies on the 64 lanes having executed the preceding LDS reads before
// any lane executes the following LDS writes.  A wavefront issues each instruction for all
// lanes at once and its LDS operations complete in issue order, so on hardware this only has
// to stop the compiler from reordering memory operations across it (no instruction emitted).
KAS_DEV void lockstep() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Identity the optimiser cannot see through.  Used on the elements of small register tables
// before a select chain: without it LLVM folds "select(i==j, a[j], ...)" back into a
// dynamically indexed load and the table moves to scratch (= global memory).
KAS_DEV int32_t opaque(int32_t v) {
  asm("" : "+v"(v));
  return v;
}

KAS_DEV int popc(uint64_t m) { return __popcll((unsigned long long)m); }

// index of the lowest set bit; m must be non-zero
KAS_DEV int first_lane(uint64_t m) { return __ffsll((unsigned long long)m) - 1; }

KAS_DEV uint64_t lanemask_lt() { return (1ull << lane()) - 1ull; }

KAS_DEV int lds_atomic_add(int* p, int v) { return atomicAdd(p, v); }

KAS_DEV void lds_atomic_or_u64(uint64_t* p, uint64_t v) {
  __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

KAS_DEV uint32_t lds_atomic_max(uint32_t* p, uint32_t v) { return atomicMax(p, v); }

// 64-bit word written earlier by this wave (accept-mask scratch): force a vector load so the
// value never comes from the scalar cache, which is not coherent with the wave's own stores.
KAS_DEV uint64_t load_shared_u64(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
KAS_DEV void store_shared_u64(uint64_t* p, uint64_t v) {
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
