// kas_hip.hip — gfx950 kernels + the C ABI of include/kas_abi.h (libkas_hip.so).
//
// A solve is three launches on one stream (device code in kas_solver_body.h):
//   kas_fill_kernel<W, NW>              P0-P4: one workgroup of NW wavefronts per scenario
//   kas_order_permutation_kernel        scenarios by descending P5 chain length
//   kas_order_relax_kernel<W,DUAL,CTX>  P5, relaxation form: one wavefront per scenario (lists <= 3 wide)
//   (or kas_order_ticket_kernel<W, G, PK>, behind kas_order_permutation_kernel: solver / stager / retirer wavefronts
//    per G scenarios; kas_order_wide_kernel<W> for lists 4 and 5 wide; kas_order_round_kernel<W>, the round form)
// Scenarios share nothing, so there is no inter-workgroup communication at all: each workgroup
// streams its own tables from HBM (coalesced 64-row tiles per wave), keeps its node state in LDS
// and writes its own out rows and one 32-byte result record.  Workgroup b lands on XCD b % 8; in
// the what-if layout (many scenarios over one shared cur table) neighbouring scenarios therefore
// spread the shared table over all eight L2s, and the 256 MiB Infinity Cache holds it once.
#define KAS_ABI_FN __host__ __device__ static inline
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kas_abi.h"
#include "kas_plan_math.h"
#include "kas_solver_body.h"

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// min waves per SIMD of the fill kernel.  Its row scans keep four tiles of rows in flight per lane
// and spill heavily below 128 VGPRs: measured in round 2 (mid rows, lookup mode hoisted, event
// counters compiled out) one batch alone takes 1.30 ms at 128 VGPRs (4 waves per SIMD, 4 workgroups
// per CU) against 1.61 ms at 96 (5, the LDS limit), and with 8 batches in flight the whole-job rate
// is the same within noise (297-299k scenarios/s), so the faster single launch wins.
#ifndef KAS_FILL_MIN_WAVES
#define KAS_FILL_MIN_WAVES 4
#endif
// A launch for flagged scenarios only (KAS_FLAG_ONLY_FLAGGED: behind the slim kernel, behind the spread fill, the wide form's second
// solve) deals them to its workgroups round-robin BY RANK among the flagged ones (kas::fill_block, kas_solver_body.h) — a tenth of
// 1000 scenarios handed back is a hundred workgroups with one scenario each, not 256 of which some hold three (round 6: 25.9 -> 15 ms
// for such a batch, scripts/handback_probe.py) — and workgroup 0 leaves their number where the plan sizes its next such launch
// (KasLaunch::handback, kas_plan_back_grid).
template <int W, int NW>
__global__ __launch_bounds__(64 * NW, KAS_FILL_MIN_WAVES) void kas_fill_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::fill_block<W, NW>(a, (int32_t)blockIdx.x, (int32_t)gridDim.x, kas_lds);
}

// The slim fill (round 6): fill_scenario<W, 4, SLIM> — only the path every BASELINE config at RF <= 3 takes (kas_solver_body.h,
// fill_topic's SLIM branch): 120 VGPRs and no scratch where kas_fill_kernel<3,4> has 128 and 368 B per lane.  A scenario that
// needs another path is flagged (KasLaunch::sp_flag) and kas_fill_kernel, launched behind this one with KAS_FLAG_ONLY_FLAGGED on
// a grid of at most KAS_FILL_BACK_GRID workgroups, solves it from its first topic.  Its LDS layout holds only what that path
// touches (kas_fill_slim_lds: 31.6 KB at 1,050 brokers against 35.5).  Measured (experiments/README.md): the smaller layout +1.4 %
// in flight; 96 VGPRs (KAS_SLIM_MIN_WAVES=5: 88 B of scratch) with five workgroups per CU -0.5 %, with four and room for two order
// wavefronts per SIMD beside them (KAS_TUNE_SLIM_LDS_PAD=1280) -2 %.
#ifndef KAS_SLIM_MIN_WAVES
#define KAS_SLIM_MIN_WAVES 4
#endif
#ifndef KAS_TUNE_SLIM_LDS_PAD
#define KAS_TUNE_SLIM_LDS_PAD 0
#endif
// (M32: the instance for launches with dword mid rows, KAS_FLAG_MID32 — one store path in either instance)
template <int W, bool M32 = false>
__global__ __launch_bounds__(256, KAS_SLIM_MIN_WAVES) void kas_fill_slim_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  for (int32_t s = (int32_t)blockIdx.x; s < a.n_scenarios; s += (int32_t)gridDim.x)
    kas::fill_scenario<W, 4, true, M32 ? 1 : 0>(a, s, kas_lds);
}
static void (*kas_fill_slim_m32())(KasLaunch) { return kas_fill_slim_kernel<3, true>; }
#define KAS_FILL_BACK_GRID 256u
#define KAS_FILL_BACK_GRID_STEP 64u

// min waves per SIMD of the ticket-form order kernel (0 = whatever the allocation comes to: 92 VGPRs, 5)
#ifndef KAS_ORDER_MIN_WAVES
#define KAS_ORDER_MIN_WAVES 0
#endif
#if KAS_ORDER_MIN_WAVES > 0
#define KAS_ORDER_BOUNDS __launch_bounds__(192, KAS_ORDER_MIN_WAVES)
#else
#define KAS_ORDER_BOUNDS __launch_bounds__(192)
#endif
template <int W, int G, bool PK>
__global__ KAS_ORDER_BOUNDS void kas_order_ticket_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::order_tickets<W, G, PK>(a, (int32_t)blockIdx.x * G, kas_lds);
}

__global__ __launch_bounds__(64 * KAS_PERM_WAVES) void kas_order_permutation_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::order_permutation(a, kas_lds);
}

template <int W>
__global__ __launch_bounds__(64) void kas_order_round_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::order_scenario_rounds<W>(a, (int32_t)blockIdx.x, kas_lds);
}

// first fit (P4) of the topics the fill kernel handed over: one workgroup of four wavefronts per scenario on a slim LDS layout
// (kas_solver_body.h, p4_scenario; KAS_FLAG_SPLIT_P4)
// ONE wavefront per scenario: its windows follow each other without a hand-over between wavefronts, so nothing in the kernel
// polls — alone that is slower (fill + first fit 1.39 ms per 1000 scenarios against 1.07 ms inside the fill workgroup, 1.09 with
// four wavefronts here), with eight batches in flight it is the fastest (672k scenarios/s in a 20-step region against 617k
// inside the fill workgroup, 620k with four wavefronts, 646k with two: experiments/README.md)
#ifndef KAS_P4_KERNEL_WAVES
#define KAS_P4_KERNEL_WAVES 1
#endif
// (M32: the instance for launches with dword mid rows, KAS_FLAG_MID32 — the layout is a constant of either instance)
template <int W, bool M32 = false>
__global__ __launch_bounds__(64 * KAS_P4_KERNEL_WAVES) void kas_p4_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  if constexpr (KAS_P4_PRIO > 0) kasw::set_priority<KAS_P4_PRIO>();
  kas::p4_scenario<W, KAS_P4_KERNEL_WAVES, false, (W == 3 && M32) ? 1 : 0>(a, (int32_t)blockIdx.x, kas_lds);
}
static void (*kas_p4_m32())(KasLaunch) { return kas_p4_kernel<3, true>; }

// Self-test of the one hardware property the relaxation form of P5 and the fill's quota draw rely on and the ISA documents do
// not state: the LDS serves the lanes of ONE ds_add_rtn instruction that name the same word in ascending lane order, so
// that the value returned to lane i is the word before the instruction plus the addends of the ACTIVE lanes below i
// (kas_wave.h, lds_add_rtn_u32).  Round 5: in the regime the kernels live in — four wavefronts per workgroup of which
// three hammer the same CU's LDS with atomics (with and without return, as the fill kernel's histogram and quota draws do),
// enough workgroups to put several on every CU at once, and every other round under a random EXEC mask (the fill draws
// its quota from inside a branch; round 4's test ran all 64 lanes always).  Pseudo-random words of tables of 1 .. 1024
// entries; *bad counts the lane-operations that came back with anything else.
__global__ __launch_bounds__(256) void kas_lds_order_selftest_kernel(unsigned int* bad, int iters) {
  __shared__ uint32_t tab[1024];
  __shared__ uint32_t noise[2048];
  const int lane = (int)(threadIdx.x & 63u);
  for (int i = (int)threadIdx.x; i < 1024; i += (int)blockDim.x) { tab[i] = 0u; noise[i] = 0u; noise[1024 + i] = 0u; }
  __syncthreads();
  uint32_t rng = 0x9E3779B9u * (blockIdx.x * 1024u + threadIdx.x + 1u);
  unsigned int nbad = 0;
  if (threadIdx.x < 64u) {
    const uint32_t mask = (1u << (2u * (blockIdx.x % 6u))) - 1u;      // 1, 4, 16, 64, 256, 1024 words
    for (int it = 0; it < iters; ++it) {
      rng = rng * 1664525u + 1013904223u;
      const uint32_t w = (rng >> 11) & mask, add = 1u + ((rng >> 5) & 15u);
      const bool on = (it & 1) == 0 || ((rng >> 24) & 3u) != 0u;      // odd rounds: a random three quarters of the lanes draw
      const uint32_t before = tab[w];                               // (only this wavefront writes tab)
      kasw::lockstep();
      uint32_t got = 0u;
      if (on) got = kasw::lds_add_rtn_u32(&tab[w], add);
      kasw::lockstep();
      const uint64_t onm = kasw::ballot(on);
      uint32_t lower = 0u;
      for (int l = 0; l < 64; ++l) {
        const uint32_t wl = (uint32_t)__builtin_amdgcn_readlane((int)w, l), al = (uint32_t)__builtin_amdgcn_readlane((int)add, l);
        lower += (l < lane && wl == w && ((onm >> l) & 1ull)) ? al : 0u;
      }
      nbad += (on && got != before + lower) ? 1u : 0u;
    }
  } else {
    for (int it = 0; it < 2 * iters; ++it) {
      rng = rng * 1664525u + 1013904223u;
      const uint32_t w = (rng >> 9) & 2047u;
      if (it & 1) atomicAdd(&noise[w], 1u << (16u * ((rng >> 3) & 1u)));       // (the fill's uint16 histogram cells)
      else if ((rng >> 28) & 1u) (void)kasw::lds_add_rtn_u32(&noise[w], 0xffffffffu);   // (its quota draw, under a mask)
    }
  }
  if (nbad) atomicAdd(bad, nbad);
}

// lists up to 3 wide: the relaxation form, one wavefront (= one workgroup) per scenario (kas_order_relax.h)
// (DUAL: the instance with double tiles, kas_relax_double_tiles)
// (CTX: the instance for batches in which some scenario hands a Context in or wants it back)
// (VERIFY: the instances for plans with KAS_PLAN_VERIFY_SAMPLE)
// (C16: the instances for plans with 16-bit cells, kas_plan_create16)
// (IDL: the instances for int32 cells with the scenario's broker ids in the LDS — kas_relax_lds_ids; the gather instances,
//  IDL = false on int32 cells, exist without the sampled verification only)
// (M32: the instances for launches with dword mid rows, KAS_FLAG_MID32 — lists 3 wide, IDL, no Context, no sampled verification)
// (QUAD: the instance with quad tiles — 256 rows a step — on dword mid rows, KAS_PLAN_RELAX_TILES(3))
template <int W, bool DUAL, bool CTX, bool VERIFY = false, bool C16 = false, bool IDL = false, bool M32 = false, bool QUAD = false>
__global__ __launch_bounds__(64) void kas_order_relax_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::order_relax<W, DUAL, CTX, VERIFY, C16, IDL, false, M32, QUAD>(a, (int32_t)blockIdx.x, kas_lds);
}
// (tiles: 0 = 64 rows a step, 1 = double tiles, 2 = quad tiles)
static void (*kas_order_relax_m32_pick(int Wc, int tiles))(KasLaunch) {
  if (Wc != 3) return nullptr;
  if (tiles >= 2) return kas_order_relax_kernel<3, true, false, false, false, true, true, true>;
  return tiles ? kas_order_relax_kernel<3, true, false, false, false, true, true> : kas_order_relax_kernel<3, false, false, false, false, true, true>;
}
template <bool VERIFY, bool C16 = false, bool IDL = false>
static void (*kas_order_relax_pick(int Wc, int dual, int ctx))(KasLaunch) {
  if (Wc <= 2) return ctx ? kas_order_relax_kernel<2, false, true, VERIFY, C16, IDL> : kas_order_relax_kernel<2, false, false, VERIFY, C16, IDL>;   // (double tiles are rows of three holders)
  if (Wc == 3) {
    if (ctx) return dual ? kas_order_relax_kernel<3, true, true, VERIFY, C16, IDL> : kas_order_relax_kernel<3, false, true, VERIFY, C16, IDL>;
    return dual ? kas_order_relax_kernel<3, true, false, VERIFY, C16, IDL> : kas_order_relax_kernel<3, false, false, VERIFY, C16, IDL>;
  }
  return nullptr;
}
// the instance for (verify, 16-bit cells, ids in the LDS)
static void (*kas_order_relax_any(int Wc, int dual, int ctx, int verify, int c16, int idl))(KasLaunch) {
  if (c16) return verify ? kas_order_relax_pick<true, true>(Wc, dual, ctx) : kas_order_relax_pick<false, true>(Wc, dual, ctx);
  if (idl) return verify ? kas_order_relax_pick<true, false, true>(Wc, dual, ctx) : kas_order_relax_pick<false, false, true>(Wc, dual, ctx);
  return verify ? nullptr : kas_order_relax_pick<false>(Wc, dual, ctx);
}

// first fit (P4) and the relaxation form of P5 in one workgroup of two wavefronts (kas_order_relax.h, p4_order_scenario): the order
// wavefront follows first fit's progress instead of waiting behind a kernel boundary — for launches whose latency is a scenario's
template <int W, bool DUAL, bool C16, bool IDL, bool M32 = false, bool QUAD = false>
__global__ __launch_bounds__(128) void kas_p4_order_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::p4_order_scenario<W, DUAL, C16, IDL, M32, QUAD>(a, (int32_t)blockIdx.x, kas_lds);
}
static void (*kas_p4_order_m32_pick(int Wc, int tiles))(KasLaunch) {
  if (Wc != 3) return nullptr;
  if (tiles >= 2) return kas_p4_order_kernel<3, true, false, true, true, true>;
  return tiles ? kas_p4_order_kernel<3, true, false, true, true> : kas_p4_order_kernel<3, false, false, true, true>;
}
// (int32 cells with the broker ids in the LDS, or 16-bit cells: the instances that wait for no gather)
static void (*kas_p4_order_pick(int Wc, int dual, int c16))(KasLaunch) {
  if (Wc <= 2) return c16 ? kas_p4_order_kernel<2, false, true, false> : kas_p4_order_kernel<2, false, false, true>;
  if (Wc == 3) {
    if (c16) return dual ? kas_p4_order_kernel<3, true, true, false> : kas_p4_order_kernel<3, false, true, false>;
    return dual ? kas_p4_order_kernel<3, true, false, true> : kas_p4_order_kernel<3, false, false, true>;
  }
  return nullptr;
}

// lists 4 and 5 wide, relaxation form: one wavefront (= one workgroup) per scenario (kas_order_relax_wide.h)
template <int W>
__global__ __launch_bounds__(64) void kas_order_relax_wide_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::order_relax_wide<W>(a, (int32_t)blockIdx.x, kas_lds);
}

// lists 4 and 5 wide: one scenario per workgroup (stager, retirer, three solver wavefronts: kas_order_wide.h)
template <int W>
__global__ __launch_bounds__(KAS_ORDER_WIDE_BLOCK) void kas_order_wide_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::order_tickets_wide<W>(a, (int32_t)blockIdx.x, kas_lds);
}

// spread fill (kas_solver_body.h, "Spread fill"): (scenario, chunk) one-wavefront workgroups for the two
// row scans, a thread per (scenario, node) for the quota, one workgroup per scenario for P4
template <int W>
__global__ __launch_bounds__(64) void kas_spread_a_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::spread_pass_a<W>(a, (int32_t)blockIdx.y, (int32_t)blockIdx.x, kas_lds);
}
template <int W>
__global__ __launch_bounds__(256) void kas_spread_q_kernel(KasLaunch a) {
  kas::spread_quota<W>(a, (int32_t)blockIdx.y, (int32_t)(blockIdx.x * 256u + threadIdx.x));
}
template <int W>
__global__ __launch_bounds__(64) void kas_spread_b_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::spread_pass_b<W>(a, (int32_t)blockIdx.y, (int32_t)blockIdx.x, kas_lds);
}
template <int W>
__global__ __launch_bounds__(64 * KAS_SPREAD_P4_WAVES) void kas_spread_p4_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  kas::spread_p4<W, KAS_SPREAD_P4_WAVES>(a, (int32_t)blockIdx.x, kas_lds);
}
struct KasSpreadKernels { void (*a)(KasLaunch); void (*q)(KasLaunch); void (*b)(KasLaunch); void (*p4)(KasLaunch); };
template <int W>
static KasSpreadKernels kas_spread_kernels_w() {
  return KasSpreadKernels{kas_spread_a_kernel<W>, kas_spread_q_kernel<W>, kas_spread_b_kernel<W>, kas_spread_p4_kernel<W>};
}

// tuning builds only: extra dynamic LDS per workgroup, to measure how much residency is worth
#ifndef KAS_TUNE_ORDER_LDS_PAD
#define KAS_TUNE_ORDER_LDS_PAD 0
#endif
#ifndef KAS_TUNE_FILL_LDS_PAD
#define KAS_TUNE_FILL_LDS_PAD 0
#endif
typedef void (*kas_kernel_fn)(KasLaunch);
#if defined(KAS_MINIMAL_INSTANCES) && KAS_MINIMAL_INSTANCES == 5
// tuning build for BASELINE.json configs[4] (lists 5 wide, 4 fill waves): seconds to compile
static bool kas_minimal_ok(int Wc, int NW, int) { return Wc == 5 && NW == 4; }
static kas_kernel_fn kas_fill_for(int, int) { return kas_fill_kernel<5, 4>; }
static kas_kernel_fn kas_fill_slim_for(int) { return nullptr; }
static kas_kernel_fn kas_p4_for(int) { return kas_p4_kernel<5>; }
static kas_kernel_fn kas_order_ticket_for(int, int, int) { return nullptr; }
static kas_kernel_fn kas_order_round_for(int) { return kas_order_round_kernel<5>; }
static kas_kernel_fn kas_order_wide_for(int) { return kas_order_wide_kernel<5>; }
static kas_kernel_fn kas_order_relaxw_for(int) { return kas_order_relax_wide_kernel<5>; }
static kas_kernel_fn kas_p4_order_for(int, int, int) { return nullptr; }
static kas_kernel_fn kas_order_relax_for(int, int, int, int = 0, int = 0, int = 0) { return nullptr; }
static KasSpreadKernels kas_spread_for(int Wc) { return Wc == 5 ? kas_spread_kernels_w<5>() : KasSpreadKernels{nullptr, nullptr, nullptr, nullptr}; }
#elif defined(KAS_MINIMAL_INSTANCES) && KAS_MINIMAL_INSTANCES != 0
// tuning builds (scripts/build_variant.sh): only the kernels BASELINE.json configs[2] launches —
// lists 3 wide, 4 fill waves, 2 scenarios per solver wavefront — so that a variant compiles in
// seconds.  Other shapes are refused by kas_plan_create in such a build.
static bool kas_minimal_ok(int Wc, int NW, int G) { return Wc == 3 && NW == 4 && (G == 2 || G == 1); }
static kas_kernel_fn kas_fill_for(int, int) { return kas_fill_kernel<3, 4>; }
static kas_kernel_fn kas_fill_slim_for(int) { return kas_fill_slim_kernel<3>; }
static kas_kernel_fn kas_p4_for(int) { return kas_p4_kernel<3>; }
static kas_kernel_fn kas_order_ticket_for(int, int G, int packed) {
  if (G == 1) return packed ? kas_order_ticket_kernel<3, 1, true> : kas_order_ticket_kernel<3, 1, false>;
  return packed ? kas_order_ticket_kernel<3, 2, true> : kas_order_ticket_kernel<3, 2, false>;
}
static kas_kernel_fn kas_order_round_for(int) { return kas_order_round_kernel<3>; }
static kas_kernel_fn kas_order_wide_for(int) { return nullptr; }
static kas_kernel_fn kas_order_relaxw_for(int) { return nullptr; }
static kas_kernel_fn kas_p4_order_for(int, int dual, int c16) { return kas_p4_order_pick(3, dual, c16); }
static kas_kernel_fn kas_order_relax_for(int, int dual, int ctx, int verify = 0, int c16 = 0, int idl = 0) {
  return kas_order_relax_any(3, dual, ctx, verify, c16, idl);
}
static KasSpreadKernels kas_spread_for(int) { return KasSpreadKernels{nullptr, nullptr, nullptr, nullptr}; }
#else
static bool kas_minimal_ok(int, int, int) { return true; }
template <int NW>
static kas_kernel_fn kas_fill_for_w(int Wc) {
  switch (Wc) {
    case 2: return kas_fill_kernel<2, NW>;
    case 3: return kas_fill_kernel<3, NW>;
    case 4: return kas_fill_kernel<4, NW>;
    case 5: return kas_fill_kernel<5, NW>;
    default: return kas_fill_kernel<8, NW>;
  }
}
static kas_kernel_fn kas_fill_for(int Wc, int NW) {
  switch (NW) {                            // (8 wavefronts per scenario: only on the CPU emulator — each
    case 1: return kas_fill_for_w<1>(Wc);   // instantiated width costs a minute of build time and 8 was
    case 2: return kas_fill_for_w<2>(Wc);   // never the faster choice on the GPU)
    default: return kas_fill_for_w<4>(Wc);
  }
}
static kas_kernel_fn kas_fill_slim_for(int Wc) {
  return Wc == 2 ? kas_fill_slim_kernel<2> : (Wc == 3 ? kas_fill_slim_kernel<3> : nullptr);
}
static kas_kernel_fn kas_p4_for(int Wc) {
  switch (Wc) {
    case 2: return kas_p4_kernel<2>;
    case 3: return kas_p4_kernel<3>;
    case 4: return kas_p4_kernel<4>;
    case 5: return kas_p4_kernel<5>;
    default: return kas_p4_kernel<8>;
  }
}
template <int G, bool PK>
static kas_kernel_fn kas_order_ticket_for_g(int Wc) {
  switch (Wc) {
    case 2: return kas_order_ticket_kernel<2, G, PK>;
    default: return kas_order_ticket_kernel<3, G, PK>;
  }
}
static kas_kernel_fn kas_order_ticket_for(int Wc, int G, int packed) {
  switch (G) {
    case 1: return packed ? kas_order_ticket_for_g<1, true>(Wc) : kas_order_ticket_for_g<1, false>(Wc);
    case 2: return packed ? kas_order_ticket_for_g<2, true>(Wc) : kas_order_ticket_for_g<2, false>(Wc);
    default: return packed ? kas_order_ticket_for_g<4, true>(Wc) : kas_order_ticket_for_g<4, false>(Wc);
  }
}
static kas_kernel_fn kas_order_round_for(int Wc) {
  switch (Wc) {
    case 2: return kas_order_round_kernel<2>;
    case 3: return kas_order_round_kernel<3>;
    case 4: return kas_order_round_kernel<4>;
    case 5: return kas_order_round_kernel<5>;
    default: return kas_order_round_kernel<8>;
  }
}
static kas_kernel_fn kas_order_wide_for(int Wc) {
  return Wc == 4 ? kas_order_wide_kernel<4> : kas_order_wide_kernel<5>;
}
static kas_kernel_fn kas_order_relaxw_for(int Wc) {
  return Wc == 4 ? kas_order_relax_wide_kernel<4> : (Wc == 5 ? kas_order_relax_wide_kernel<5> : nullptr);
}
static kas_kernel_fn kas_p4_order_for(int Wc, int dual, int c16) { return kas_p4_order_pick(Wc, dual, c16); }
static kas_kernel_fn kas_order_relax_for(int Wc, int dual, int ctx, int verify = 0, int c16 = 0, int idl = 0) {
  return kas_order_relax_any(Wc, dual, ctx, verify, c16, idl);
}
static KasSpreadKernels kas_spread_for(int Wc) {
  switch (Wc) {
    case 3: return kas_spread_kernels_w<3>();
    case 4: return kas_spread_kernels_w<4>();
    case 5: return kas_spread_kernels_w<5>();
    default: return KasSpreadKernels{nullptr, nullptr, nullptr, nullptr};
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int set_error(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define KAS_HIP_TRY(expr)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return set_error(KAS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)

// device buffer that only ever grows (plans rebuilt in place, the host path's table buffers)
struct KasBuf { void* p = nullptr; size_t cap = 0; };

// Host-path cache of a context (kas_solve_host): device buffers that only ever grow, and the plans
// of the most recent batch shapes — a caller that solves the same cluster shape again (the CLI's
// per-topic loop, a JVM calling once per topic) finds its plan byte for byte; a what-if planner that
// changes the broker sets on every call gets the least recently used plan rebuilt in place over the
// scratch it already owns.  Neither pays hipMalloc / hipFree after the first call of a shape.
struct KasCachedPlan {
  kas_plan* plan = nullptr;
  uint64_t key = 0;
  std::vector<unsigned char> desc;      // the bytes the key was computed from (compared on a hit)
  uint64_t sig = 0;                     // batch size + topic descriptors only: what-if variants of one snapshot share it
  uint64_t last_use = 0;
  uint64_t call = 0;                    // host call that last used the entry (its ranges must not evict each other)
};
#define KAS_HOST_PLAN_CACHE 16
#ifndef KAS_HOST_STREAMS
#define KAS_HOST_STREAMS 8   // (round 4: 3 -> 8, a chain per scenario range: 11.0k -> 12.9k scenarios/s through kas_solve_host, gpurun_out/r4n)
#endif
struct kas_ctx {
  int device;
  hipStream_t stream;
  hipStream_t hstream[KAS_HOST_STREAMS];   // the host path's chains (upload -> solve -> download of a scenario range)
  hipEvent_t hevent;                       // "shared pools are up" of the current host call
  // a call cut into scenario ranges: every upload on one stream, every download on another (copies of one direction
  // queue behind each other anyway, and the two directions only run at the same time when no stream carries both),
  // the solves on hstream[]; events hand a range from upload to solve to download
  hipStream_t hup, hdown;
  hipEvent_t hev_up[KAS_HOST_STREAMS], hev_done[KAS_HOST_STREAMS];
  std::mutex host_mu;                   // kas_solve_host calls on one context are serialised
  KasBuf h_cur, h_out, h_aux, h_ctx, h_tr, h_sr;
  KasBuf h_cur16, h_out16;              // the 16-bit cells of kas_solve_host16 as they travel (widened / narrowed on the device)
  std::vector<int32_t> ident_ids;       // node_id pool of a 16-bit call: node i of every scenario has id i
  uint64_t ident_stamp = 0;             // ... and which node ranges it was filled for (kas_ident_batch)
  KasBuf h_tr_pin, h_sr_pin;            // pinned HOST staging of the result records (see kas_solve_host_locked)
  KasCachedPlan plans[KAS_HOST_PLAN_CACHE];
  uint64_t use_clock = 0;
  uint64_t host_calls = 0, host_plan_hits = 0, host_allocs = 0;
  int lds_lane_order_ok = 0;            // the self-test passed: the relaxation form of P5 and the fill's quota draw by ds_add_rtn may run here
  int lds_lane_order_state = -1;        // 1 passed, 0 FAILED (some lane-operation came back out of lane order), -1 could not run
                                        // (allocation / launch error), -2 switched off (environment: KAS_NO_LANE_ORDER=1)
  long lds_lane_order_checked = 0;      // lane-operations checked so far (context creation + every plan creation)
};

#define KAS_TIMER_SLOTS 64

struct kas_plan {
  uint32_t full_fill = 0;               // KAS_PLAN_FULL_FILL of the last kas_plan_set_flags: no slim fill kernel
  uint32_t mid32_bits = 0;              // KAS_PLAN_NO_MID32 / KAS_PLAN_MID32 of the last kas_plan_set_flags (kas_mid32_wanted)
  uint32_t index_rows_bits = 0;         // KAS_PLAN_NO_INDEX_ROWS / KAS_PLAN_INDEX_ROWS of the last kas_plan_set_flags (kas_index_rows_wanted)
  kas_ctx* ctx;
  KasShape shape;
  int Wc;                       // instantiated width class
  int NW;                       // wavefronts per scenario workgroup of the fill kernel
  int G;                        // scenarios per wavefront of the ticket-form order kernel
  int tickets;                  // 1: ticket form of P5, 0: round form
  int fused;                    // 1: per-chunk histograms in the fill (KasShape::fused_ok), see kas_plan_fused()
  uint32_t flags;               // KAS_FLAG_*
  KasLds lds, lds_fused;
  int32_t n_scenarios, n_topics;
  // device copies and scratch owned by the plan (grow-only: a plan can be rebuilt for another batch)
  KasBuf b_scen, b_topics, b_node_id, b_node_rack, b_accmask_off, b_accmask, b_orph_off, b_orph, b_perm, b_stats,
         b_ord_flag, b_sp_hist, b_sp_quota, b_sp_node, b_sp_flag, b_sp_oc, b_p4s;
  uint64_t* allocs;             // allocation counter to report to (the context's, or NULL)
  int single_topic;             // every scenario has exactly one topic
  int cells16;                  // kas_plan_create16: cur / out cells are uint16 node indices (KAS_FLAG_CELLS16 in every launch)
  int32_t sp_alloc_chunks;      // chunks per scenario the spread-fill scratch is sized for
  int32_t* h_handback = nullptr; // pinned host word the device writes: scenarios the slim fill kernel handed back in the last solve that has got that far
  int32_t* d_handback = nullptr; // ... its device address
  hipStream_t last_stream;
  int last_slot;                // timer slot of the most recent solve (-1: none yet)
  // kernel timing: event pairs recorded around every launch on the launch stream
  hipEvent_t ev_start[KAS_TIMER_SLOTS], ev_mid[KAS_TIMER_SLOTS], ev_stop[KAS_TIMER_SLOTS];
  int timer_next, timer_count;
};

static int kas_buf_reserve(KasBuf* b, size_t bytes, uint64_t* allocs, const char* what) {
  if (bytes == 0) bytes = 16;
  if (bytes <= b->cap) return KAS_E_OK;
  if (b->p) { (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
  hipError_t e = hipMalloc(&b->p, bytes);
  if (e != hipSuccess) { b->p = nullptr; return set_error(KAS_E_NOMEM, std::string(what) + ": " + hipGetErrorString(e)); }
  b->cap = bytes;
  if (allocs) *allocs += 1;
  return KAS_E_OK;
}
static void kas_buf_free(KasBuf* b) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }

#pragma GCC visibility push(default)
extern "C" {

int kas_abi_version(void) { return KAS_ABI_VERSION; }

const char* kas_strerror(int code) {
  switch (code) {
    case KAS_E_OK: return "ok";
    case KAS_E_INVALID_ARG: return "invalid argument";
    case KAS_E_HIP: return "HIP runtime error or no HIP device";
    case KAS_E_UNSUPPORTED: return "batch shape not supported by the kernels";
    case KAS_E_NOMEM: return "out of memory";
    default: return "unknown error code";
  }
}

const char* kas_status_string(int status) {
  switch (status) {
    case KAS_OK: return "OK";
    case KAS_FAIL_UNASSIGNABLE: return "partition could not be fully assigned (KAS:183-184)";
    case KAS_FAIL_RF_NOT_POSITIVE: return "replication factor is not positive (KTA:65-66)";
    case KAS_FAIL_RF_GT_BROKERS: return "replication factor exceeds available brokers (KTA:67-69)";
    case KAS_FAIL_HASH_INDEX: return "topic hashCode is Integer.MIN_VALUE: negative index (KAS:190-192)";
    case KAS_FAIL_RF_MISMATCH: return "partition with unexpected replication factor (KTA:58-60)";
    case KAS_SKIPPED: return "skipped: an earlier topic of the scenario failed";
    case KAS_FAIL_BAD_NODES: return "node table not strictly ascending / non-negative, or rack out of range";
    case KAS_FAIL_WATCHDOG: return "a wavefront polled past KAS_SPIN_BOUND without progress (internal error: the solve was abandoned instead of hanging)";
    default: return "unknown status";
  }
}

const char* kas_last_error(void) { return g_last_error.c_str(); }

// KTA:47-69 (host arithmetic): see include/kas_abi.h
int kas_resolve_replication_factor(const int32_t* partition_ids, const int32_t* list_sizes, int32_t n_partitions,
                                   int32_t desired_rf, int32_t n_brokers, kas_rf_result* res) {
  if (!res || n_partitions < 0 || (n_partitions > 0 && (!partition_ids || !list_sizes)))
    return set_error(KAS_E_INVALID_ARG, "kas_resolve_replication_factor: null argument or negative count");
  res->status = KAS_OK; res->fail_partition = -1; res->fail_list_size = -1;
  int32_t rf = desired_rf;
  for (int32_t i = 0; i < n_partitions; ++i) {
    if (rf < 0) {
      rf = list_sizes[i];                                      // KTA:55-56 (a list cannot be shorter than empty)
    } else if (desired_rf < 0 && rf != list_sizes[i]) {        // KTA:57-60
      res->status = KAS_FAIL_RF_MISMATCH; res->rf = rf;
      res->fail_partition = partition_ids[i]; res->fail_list_size = list_sizes[i];
      return 0;
    }
  }
  res->rf = rf;
  if (!(rf > 0)) res->status = KAS_FAIL_RF_NOT_POSITIVE;       // KTA:65-66
  else if (!(rf <= n_brokers)) res->status = KAS_FAIL_RF_GT_BROKERS;   // KTA:67-69
  return 0;
}

int kas_failure_text(const char* topic, int32_t status, int32_t fail_partition, int32_t rf, int32_t list_size, char* buf, int n) {
  if (!buf || n <= 0) return 0;
  buf[0] = 0;
  const std::string t = topic ? topic : "null";               // (Java prints a null String as "null")
  std::string m;
  switch (status) {
    case KAS_FAIL_UNASSIGNABLE: m = "Partition " + std::to_string(fail_partition) + " could not be fully assigned!"; break;
    case KAS_FAIL_RF_MISMATCH:
      m = "Topic " + t + " has partition " + std::to_string(fail_partition) + " with unexpected replication factor " + std::to_string(list_size);
      break;
    case KAS_FAIL_RF_NOT_POSITIVE: m = "Topic " + t + " does not have a positive replication factor!"; break;
    case KAS_FAIL_RF_GT_BROKERS: m = "Topic " + t + " has a higher replication factor (" + std::to_string(rf) + ") than available brokers!"; break;
    default: return 0;
  }
  const size_t len = m.size() < (size_t)(n - 1) ? m.size() : (size_t)(n - 1);
  memcpy(buf, m.data(), len);
  buf[len] = 0;
  return (int)len;
}

int kas_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Runs the self-test (grid workgroups x iters rounds x 64 lanes) on the context's stream, blocking; folds the outcome into
// the context: a failure is final, a run that could not happen leaves an earlier pass standing.
static void kas_lds_order_selftest(kas_ctx* c, int grid, int iters) {
  if (c->lds_lane_order_state == 0 || c->lds_lane_order_state == -2) return;
  if (const char* off = getenv("KAS_NO_LANE_ORDER")) {
    if (off[0] != 0 && off[0] != '0') { c->lds_lane_order_state = -2; c->lds_lane_order_ok = 0; return; }
  }
  unsigned int* d_bad = nullptr;
  unsigned int h_bad = 1u;
  bool ran = false;
  if (hipMalloc((void**)&d_bad, sizeof(unsigned int)) == hipSuccess) {
    if (hipMemsetAsync(d_bad, 0, sizeof(unsigned int), c->stream) == hipSuccess) {
      hipLaunchKernelGGL(kas_lds_order_selftest_kernel, dim3((unsigned)grid), dim3(256), 0, c->stream, d_bad, iters);
      ran = hipGetLastError() == hipSuccess &&
            hipMemcpyAsync(&h_bad, d_bad, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
            hipStreamSynchronize(c->stream) == hipSuccess;
    }
    (void)hipFree(d_bad);
  }
  if (!ran) { (void)hipGetLastError(); return; }             // (state stays: -1 if it never ran, 1 if an earlier run passed)
  c->lds_lane_order_checked += (long)grid * iters * 64;
  c->lds_lane_order_state = h_bad == 0u ? 1 : 0;
  c->lds_lane_order_ok = h_bad == 0u ? 1 : 0;
}

int kas_ctx_create(int device, kas_ctx** out_ctx) {
  if (!out_ctx) return set_error(KAS_E_INVALID_ARG, "out_ctx == NULL");
  *out_ctx = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(KAS_E_HIP, "no HIP device visible: this library has no CPU path");
  if (device < 0 || device >= n) return set_error(KAS_E_INVALID_ARG, "device index out of range");
  KAS_HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  KAS_HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return set_error(KAS_E_HIP, std::string("device is ") + prop.gcnArchName +
                                    ", the kernels are built for gfx950 only");
  kas_ctx* c = new kas_ctx();
  c->device = device;
  c->stream = nullptr; c->hevent = nullptr; c->hup = nullptr; c->hdown = nullptr;
  for (hipStream_t& h : c->hstream) h = nullptr;
  for (hipEvent_t& ev : c->hev_up) ev = nullptr;
  for (hipEvent_t& ev : c->hev_done) ev = nullptr;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  // (hstream[]: made by the first host call that needs them, kas_host_solve_streams — round 5: a context used to create
  // all eight with itself, each took one of the process's GPU_MAX_HW_QUEUES hardware queues, and the streams of a caller
  // that never makes a host call — bench.py's eight slots — were left to share what remained: the kernel trace showed
  // two pairs of slots on one queue each, their solves serialised.  hup / hdown: kas_host_copy_streams.)
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->hevent, hipEventDisableTiming);
  for (hipEvent_t& ev : c->hev_up) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  for (hipEvent_t& ev : c->hev_done) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    kas_ctx_destroy(c);
    return set_error(KAS_E_HIP, hipGetErrorString(e));
  }
  // the relaxation form of P5 runs only where the LDS hands out the lanes' additions in lane order (see the kernel)
  kas_lds_order_selftest(c, 2048, 200);
  *out_ctx = c;
  return KAS_E_OK;
}

void kas_ctx_destroy(kas_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (hipStream_t h : ctx->hstream) if (h) (void)hipStreamSynchronize(h);
  if (ctx->hup) (void)hipStreamSynchronize(ctx->hup);
  if (ctx->hdown) (void)hipStreamSynchronize(ctx->hdown);
  for (KasCachedPlan& c : ctx->plans) if (c.plan) kas_plan_destroy(c.plan);
  for (KasBuf* b : {&ctx->h_cur, &ctx->h_out, &ctx->h_aux, &ctx->h_ctx, &ctx->h_tr, &ctx->h_sr, &ctx->h_cur16, &ctx->h_out16}) kas_buf_free(b);
  for (KasBuf* b : {&ctx->h_tr_pin, &ctx->h_sr_pin}) { if (b->p) (void)hipHostFree(b->p); b->p = nullptr; b->cap = 0; }
  if (ctx->hevent) (void)hipEventDestroy(ctx->hevent);
  for (hipEvent_t ev : ctx->hev_up) if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->hev_done) if (ev) (void)hipEventDestroy(ev);
  if (ctx->hup) (void)hipStreamDestroy(ctx->hup);
  if (ctx->hdown) (void)hipStreamDestroy(ctx->hdown);
  for (hipStream_t h : ctx->hstream) if (h) (void)hipStreamDestroy(h);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int kas_ctx_lds_lane_order(const kas_ctx* ctx, int64_t* lane_ops_checked) {
  if (!ctx) return set_error(KAS_E_INVALID_ARG, "ctx == NULL");
  if (lane_ops_checked) *lane_ops_checked = (int64_t)ctx->lds_lane_order_checked;
  return ctx->lds_lane_order_state;
}

int kas_ctx_synchronize(kas_ctx* ctx) {
  if (!ctx) return set_error(KAS_E_INVALID_ARG, "ctx == NULL");
  KAS_HIP_TRY(hipSetDevice(ctx->device));
  KAS_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return KAS_E_OK;
}

void kas_plan_destroy(kas_plan* p) {
  if (!p) return;
  (void)hipSetDevice(p->ctx->device);
  (void)hipStreamSynchronize(p->ctx->stream);
  if (p->last_slot >= 0) (void)hipEventSynchronize(p->ev_stop[p->last_slot]);
  for (KasBuf* b : {&p->b_scen, &p->b_topics, &p->b_node_id, &p->b_node_rack, &p->b_accmask_off, &p->b_accmask,
                    &p->b_orph_off, &p->b_orph, &p->b_perm, &p->b_stats, &p->b_ord_flag, &p->b_sp_hist, &p->b_sp_quota,
                    &p->b_sp_node, &p->b_sp_flag, &p->b_sp_oc, &p->b_p4s})
    kas_buf_free(b);
  for (int i = 0; i < KAS_TIMER_SLOTS; ++i) {
    if (p->ev_start[i]) (void)hipEventDestroy(p->ev_start[i]);
    if (p->ev_stop[i]) (void)hipEventDestroy(p->ev_stop[i]);
    if (p->ev_mid[i]) (void)hipEventDestroy(p->ev_mid[i]);
  }
  if (p->h_handback) (void)hipHostFree(p->h_handback);
  delete p;
}

// the relaxation form's instances for this plan keep the broker ids in the LDS (int32 cells; kas_relax_lds_ids)
static int kas_plan_relax_idl(const kas_plan* p) {
  return !p->cells16 && kas_relax_lds_ids(p->shape.n_max, p->shape.any_ctx) ? 1 : 0;
}

// opt every kernel this plan may launch into its dynamic LDS size
static int kas_plan_set_kernels(kas_plan* p) {
  KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_fill_for(p->Wc, p->NW), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (p->fused && p->lds_fused.total > p->lds.total ? p->lds_fused.total : p->lds.total) + KAS_TUNE_FILL_LDS_PAD));
  if (p->NW == 4 && p->fused && kas_fill_slim_for(p->Wc) != nullptr)
    KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_fill_slim_for(p->Wc), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kas_fill_slim_lds(p->shape.n_max, p->Wc, p->shape.idmap_entries).total + KAS_TUNE_SLIM_LDS_PAD));
  if (p->NW == 4 && p->fused && p->Wc == 3 && kas_fill_slim_for(p->Wc) != nullptr)     // (its instance for dword mid rows)
    KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_fill_slim_m32(), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kas_fill_slim_lds(p->shape.n_max, p->Wc, p->shape.idmap_entries).total + KAS_TUNE_SLIM_LDS_PAD));
  if (p->Wc <= 3 && p->tickets && kas_order_ticket_for(p->Wc, p->G, 0))   // (beyond 8,191 brokers only the relaxation form applies)
    for (int pk = 0; pk < 2; ++pk)
      KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_order_ticket_for(p->Wc, p->G, pk),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kas_order_ticket_lds(p->shape.n_max, p->G, pk) + KAS_TUNE_ORDER_LDS_PAD));
  for (int dual = 0; dual < 2; ++dual)
    for (int verify = 0; verify < 2; ++verify)
      if (p->shape.relax_ok && kas_order_relax_for(p->Wc, dual, p->shape.any_ctx, verify, p->cells16, kas_plan_relax_idl(p)))
        KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_order_relax_for(p->Wc, dual, p->shape.any_ctx, verify, p->cells16, kas_plan_relax_idl(p)),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kas_order_relax_lds(p->shape.n_max, dual, p->shape.any_ctx, kas_plan_relax_idl(p))));
  if (p->shape.with_x && kas_p4_lds_layout(p->shape.n_max).total <= KAS_LDS_LIMIT)
  {
    KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_p4_for(p->Wc), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kas_p4_lds_layout(p->shape.n_max).total));
    if (p->Wc == 3)
      KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_p4_m32(), hipFuncAttributeMaxDynamicSharedMemorySize, kas_p4_lds_layout(p->shape.n_max).total));
  }
  if (p->shape.round_fits)
    KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_order_round_for(p->Wc),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kas_order_round_lds(p->shape.n_max, p->Wc)));
  for (int dual = 0; dual < 2; ++dual)
    if (p->shape.relax_ok && !p->shape.any_ctx && (p->cells16 || kas_plan_relax_idl(p)) && kas_p4_order_for(p->Wc, dual, p->cells16) &&
        kas_p4_order_lds(p->shape.n_max, dual, kas_plan_relax_idl(p)) <= KAS_LDS_LIMIT)
      KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_p4_order_for(p->Wc, dual, p->cells16), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kas_p4_order_lds(p->shape.n_max, dual, kas_plan_relax_idl(p))));
  if (p->shape.relax_ok && !p->shape.any_ctx && kas_plan_relax_idl(p) && p->shape.n_max <= KAS_MID32_N_MAX)   // (the instances for dword mid rows)
    for (int dual = 0; dual < 3; ++dual) {                     // (tiles of 64 rows, double tiles, quad tiles)
      if (kas_order_relax_lds(p->shape.n_max, dual, 0, 1) > KAS_LDS_LIMIT) continue;
      if (kas_order_relax_m32_pick(p->Wc, dual))
        KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_order_relax_m32_pick(p->Wc, dual), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        kas_order_relax_lds(p->shape.n_max, dual, 0, 1)));
      if (kas_p4_order_m32_pick(p->Wc, dual) && kas_p4_order_lds(p->shape.n_max, dual, 1) <= KAS_LDS_LIMIT)
        KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_p4_order_m32_pick(p->Wc, dual), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        kas_p4_order_lds(p->shape.n_max, dual, 1)));
    }
  if (p->shape.relaxw_ok && kas_order_relaxw_for(p->Wc))
    KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_order_relaxw_for(p->Wc), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kas_order_relaxw_lds(p->shape.n_max, p->Wc)));
  if (p->shape.wide_ok && kas_order_wide_for(p->Wc))
    KAS_HIP_TRY(hipFuncSetAttribute((const void*)kas_order_wide_for(p->Wc),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kas_order_wide_lds(p->shape.n_max)));
  const KasSpreadKernels sk = kas_spread_for(p->Wc);
  if (sk.a && p->shape.with_x) {
    const int l1 = kas_fill_lds_layout(p->shape.n_max, p->Wc, 1, p->shape.idmap_entries, p->shape.need_bsearch, 1).total;
    const int l4 = kas_fill_lds_layout(p->shape.n_max, p->Wc, 4, p->shape.idmap_entries, p->shape.need_bsearch, 1).total;
    if (l1 <= KAS_LDS_LIMIT && l4 <= KAS_LDS_LIMIT) {
      KAS_HIP_TRY(hipFuncSetAttribute((const void*)sk.a, hipFuncAttributeMaxDynamicSharedMemorySize, l1));
      KAS_HIP_TRY(hipFuncSetAttribute((const void*)sk.b, hipFuncAttributeMaxDynamicSharedMemorySize, l1));
      KAS_HIP_TRY(hipFuncSetAttribute((const void*)sk.p4, hipFuncAttributeMaxDynamicSharedMemorySize, l4));
    }
  }
  return KAS_E_OK;
}

// per-chunk histograms: what the shape allows unless switched off (or the general fill is forced)
static bool kas_plan_fused(const kas_plan* p) {
  return p->fused && !(p->flags & (KAS_FLAG_TWO_PASS_HIST | KAS_FLAG_GENERIC_FILL));
}

// index rows in this plan's next solve (KAS_FLAG_INDEX_ROWS; the kernel still decides per topic: rows of the batch's width, a
// direct id table): int32 cells, lists up to 3 wide, per-chunk histograms, the quota drawn with the atomic-with-return
static bool kas_plan_index_rows(const kas_plan* p) {
  return !p->cells16 && kas_index_rows_wanted(p->index_rows_bits) && p->Wc <= 3 && kas_plan_fused(p) && p->ctx->lds_lane_order_ok &&
         !(p->flags & KAS_FLAG_NO_RTN_QUOTA) && p->shape.n_max < 0x3fff && p->shape.idmap_entries > 0;
}

// the slim fill kernel in this plan's next solve (with kas_fill_kernel behind it for the scenarios it hands back): int32 cells,
// lists up to 3 wide, per-chunk histograms on 4 wavefronts, the quota drawn with the atomic-with-return, a direct id table for
// every scenario, no index rows, first fit handed over (split_p4: to kas_p4_kernel or to kas_p4_order_kernel), no spread fill
static bool kas_plan_slim_fill(const kas_plan* p, bool split_p4, int32_t chunks) {
  return KAS_SLIM_FILL_DEFAULT && !p->full_fill && !p->cells16 && p->Wc <= 3 && p->NW == 4 && kas_fill_slim_for(p->Wc) != nullptr &&
         kas_plan_fused(p) && p->shape.with_x && p->ctx->lds_lane_order_ok && !(p->flags & KAS_FLAG_NO_RTN_QUOTA) &&
         !kas_plan_index_rows(p) && split_p4 && chunks == 0 && p->shape.idmap_entries > 0 && !p->shape.need_bsearch &&
         p->b_sp_flag.p != nullptr;
}

// workgroups of the kas_fill_kernel launch behind the slim kernel: KAS_FILL_BACK_GRID — each needs a 35 KB / 4 x 128-VGPR slot before it
// can see that there is nothing to do — unless the plan's last solve handed more scenarios back than that: then one workgroup per
// such scenario and a quarter more (a batch whose rows are not rack-diverse hands EVERY scenario back, and the general fill of a
// scenario is one long chain: 1000 of them on 256 workgroups took 53 ms, on 1000 they take 20).  A rebuilt plan starts small again.
static unsigned kas_plan_back_grid(const kas_plan* p, unsigned fill_grid) {
  unsigned g = KAS_FILL_BACK_GRID;
  const int32_t last = p->h_handback ? *(volatile int32_t*)p->h_handback : 0;
  if (last > 0 && (unsigned)last + (unsigned)last / 4u > g)
    g = (((unsigned)last + (unsigned)last / 4u + KAS_FILL_BACK_GRID_STEP - 1u) / KAS_FILL_BACK_GRID_STEP) * KAS_FILL_BACK_GRID_STEP;
  return g < fill_grid ? g : fill_grid;
}

// chunks per scenario of the spread fill for this plan's next solve, or 0 (one-workgroup fill kernel)
static int32_t kas_plan_spread_chunks(const kas_plan* p) {
  if (p->cells16) return 0;                                  // (the spread fill's kernels read int32 cells)
  if (p->NW != 4 || !kas_spread_for(p->Wc).a || (p->flags & KAS_FLAG_GENERIC_FILL)) return 0;
  // (the quota kernel puts scenarios on grid.y and nodes on grid.x)
  if (p->shape.n_max <= 0 || p->n_scenarios > 65535) return 0;
  if (kas_fill_lds_layout(p->shape.n_max, p->Wc, 4, p->shape.idmap_entries, p->shape.need_bsearch, 1).total > KAS_LDS_LIMIT) return 0;
  return kas_spread_chunks(p->shape, p->n_scenarios, p->single_topic != 0, (p->flags & KAS_FLAG_SPREAD_FILL) != 0);
}

// scratch of the spread fill for the plan's current flags (kas_plan_create / kas_plan_set_flags, never a solve)
static int kas_plan_spread_scratch(kas_plan* p) {
  const int32_t chunks = kas_plan_spread_chunks(p);
  if (chunks > 0) {
    const size_t S = (size_t)p->n_scenarios, NM = (size_t)(p->shape.n_max > 0 ? p->shape.n_max : 1), C = (size_t)chunks;
    int rc;
    if ((rc = kas_buf_reserve(&p->b_sp_hist, 4 * S * C * (size_t)p->Wc * NM, p->allocs, "spread-fill scratch")) != KAS_E_OK ||
        (rc = kas_buf_reserve(&p->b_sp_quota, 4 * S * C * NM, p->allocs, "spread-fill scratch")) != KAS_E_OK ||
        (rc = kas_buf_reserve(&p->b_sp_node, 4 * S * 2 * NM, p->allocs, "spread-fill scratch")) != KAS_E_OK ||
        (rc = kas_buf_reserve(&p->b_sp_flag, 4 * (S + 1), p->allocs, "spread-fill scratch")) != KAS_E_OK ||
        (rc = kas_buf_reserve(&p->b_sp_oc, 4 * S * (C + 2), p->allocs, "spread-fill scratch")) != KAS_E_OK)
      return rc;
  }
  else {                                                         // (the slim fill kernel's hand-back flags: 4 B per scenario)
    const int rc = kas_buf_reserve(&p->b_sp_flag, 4 * ((size_t)p->n_scenarios + 1), p->allocs, "fill hand-back flags");
    if (rc != KAS_E_OK) return rc;
  }
  p->sp_alloc_chunks = chunks;
  return KAS_E_OK;
}

// (Re)build a plan for a batch: shape, device copies of the descriptors and node tables, scratch.  The
// plan's buffers only grow, so rebuilding for a batch of the same size allocates nothing.
static int kas_plan_build(kas_plan* p, const kas_batch_desc* batch) {
  KasShape sh;
  std::string err;
  int rc = kas_shape_batch(batch, &sh, &err, 0, 0);
  if (rc != KAS_E_OK) return set_error(rc, err);
  kas_ctx* ctx = p->ctx;
  KAS_HIP_TRY(hipSetDevice(ctx->device));
  // a rebuilt plan may still have its last solve in flight on some stream
  if (p->last_slot >= 0) KAS_HIP_TRY(hipEventSynchronize(p->ev_stop[p->last_slot]));
  p->shape = sh;
  p->Wc = sh.Wc; p->NW = sh.NW; p->G = sh.G;
  p->tickets = sh.tickets_ok; p->fused = sh.fused_ok;
  p->lds = sh.lds; p->lds_fused = sh.lds_fused;
  p->flags = 0; p->index_rows_bits = 0; p->full_fill = 0; p->mid32_bits = 0;
  // (the hand-back count of the plan's last solve stays when the plan is rebuilt in place for another batch — the host path's
  // cached plans: a what-if caller hands over fresh broker sets on every call, and whether its rows are rack-diverse is the
  // SNAPSHOT's property.  A stale count costs one launch with idle workgroups, then it is this batch's own.)
  p->n_scenarios = batch->n_scenarios; p->n_topics = batch->n_topics;
  p->single_topic = kas_batch_single_topic(batch) ? 1 : 0;
  p->sp_alloc_chunks = 0;
  p->last_stream = ctx->stream; p->last_slot = -1;
  p->timer_next = 0; p->timer_count = 0;
  if (p->lds.total > KAS_LDS_LIMIT)
    return set_error(KAS_E_UNSUPPORTED, "LDS carve-up exceeds 160 KiB at the instantiated width");
  if (!kas_minimal_ok(p->Wc, p->NW, p->G))
    return set_error(KAS_E_UNSUPPORTED, "tuning build (KAS_MINIMAL_INSTANCES): only lists 3 wide, 4 fill waves");
  // some order kernel must be able to take the batch HERE: beyond 8,191 brokers at lists <= 3 wide that is the relaxation
  // form alone, which needs the LDS lane order the context's self-test looks for (such a plan used to fail at its
  // first solve with a launch error)
  if (p->cells16) {
    // 16-bit cells (kas_plan_create16): the kernels with that I/O are the fill kernel (+ kas_p4_kernel), the relaxation form
    // and the round form — lists up to 3 wide; anything else is the caller's to widen (kas_solve_host16 does)
    const bool relax16 = sh.relax_ok && ctx->lds_lane_order_ok && kas_order_relax_for(sh.Wc, 0, 0, 0, 1) != nullptr;
    if (sh.Wc > 3 || !(relax16 || sh.round_fits))
      return set_error(KAS_E_UNSUPPORTED, "16-bit cells: lists up to 3 wide, and a batch the relaxation form (LDS lane-order self-test passed) or the round form of the order kernel takes");
  }
  {
    const bool relax_here = sh.relax_ok && ctx->lds_lane_order_ok && kas_order_relax_for(sh.Wc, 0, 0) != nullptr;
    if (!sh.round_fits && !sh.tickets_ok && !sh.wide_ok && !relax_here)
      return set_error(KAS_E_UNSUPPORTED,
                       sh.relax_ok ? (ctx->lds_lane_order_state == -1
                                          ? "this broker count x list width is served by the relaxation form only, and the context's LDS lane-order self-test could not run"
                                          : "this broker count x list width is served by the relaxation form only, and the context does not use it (LDS lane-order self-test failed or KAS_NO_LANE_ORDER)")
                                   : "no order kernel fits this broker count x list width (INTEGRATION.md, limits)");
  }
  hipStream_t st = ctx->stream;
  const size_t S = (size_t)batch->n_scenarios, T = (size_t)batch->n_topics, NP = (size_t)batch->node_pool_len;
  struct Up { KasBuf* b; const void* src; size_t bytes; const char* what; };
  const Up ups[] = {
      {&p->b_scen, batch->scenarios, sizeof(kas_scenario_desc) * S, "scenario descriptors"},
      {&p->b_topics, batch->topics, sizeof(kas_topic_desc) * T, "topic descriptors"},
      {&p->b_node_id, batch->node_id, sizeof(int32_t) * NP, "node ids"},
      {&p->b_node_rack, batch->node_rack, sizeof(int32_t) * NP, "node racks"},
      {&p->b_accmask_off, sh.accmask_off.data(), sizeof(int64_t) * sh.accmask_off.size(), "accept-mask offsets"},
      {&p->b_orph_off, sh.orph_off.data(), sizeof(int64_t) * sh.orph_off.size(), "orphan-list offsets"},
  };
  for (const Up& u : ups) {
    if ((rc = kas_buf_reserve(u.b, u.bytes, p->allocs, u.what)) != KAS_E_OK) return rc;
    if (u.bytes > 0) KAS_HIP_TRY(hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, st));
  }
  const size_t stats_bytes = sizeof(int64_t) * KAS_STATS_PER_SCENARIO * (S + 1);
  if ((rc = kas_buf_reserve(&p->b_accmask, sizeof(uint64_t) * (size_t)(sh.accmask_words + 1), p->allocs, "accept-mask scratch")) != KAS_E_OK ||
      (rc = kas_buf_reserve(&p->b_orph, sizeof(int32_t) * (size_t)(sh.orph_ints + 64), p->allocs, "orphan-list scratch")) != KAS_E_OK ||
      (rc = kas_buf_reserve(&p->b_perm, sizeof(int32_t) * (S + 1), p->allocs, "scenario-order scratch")) != KAS_E_OK ||
      (rc = kas_buf_reserve(&p->b_ord_flag, sizeof(int32_t) * (S + 1), p->allocs, "order-form flags")) != KAS_E_OK ||
      (rc = kas_buf_reserve(&p->b_stats, stats_bytes, p->allocs, "stats buffer")) != KAS_E_OK)
    return rc;
  KAS_HIP_TRY(hipMemsetAsync(p->b_stats.p, 0, stats_bytes, st));
  // hand-over of the first fit to kas_p4_kernel: head words + the brokers' loads, per topic
  if (sh.with_x && (rc = kas_buf_reserve(&p->b_p4s, sizeof(int32_t) * T * (size_t)(KAS_P4S_HEAD + (sh.n_max > 0 ? sh.n_max : 1)),
                                         p->allocs, "first-fit hand-over scratch")) != KAS_E_OK)
    return rc;
  if ((rc = kas_plan_spread_scratch(p)) != KAS_E_OK) return rc;
  if ((rc = kas_plan_set_kernels(p)) != KAS_E_OK) return rc;
  // descriptors are resident before the caller may free its copies
  KAS_HIP_TRY(hipStreamSynchronize(st));
  return KAS_E_OK;
}

static int kas_plan_new(kas_ctx* ctx, const kas_batch_desc* batch, uint64_t* allocs, kas_plan** out_plan, int cells16 = 0) {
  *out_plan = nullptr;
  kas_plan* p = new kas_plan();
  p->cells16 = cells16;
  memset((void*)p->ev_start, 0, sizeof(p->ev_start));
  memset((void*)p->ev_stop, 0, sizeof(p->ev_stop));
  memset((void*)p->ev_mid, 0, sizeof(p->ev_mid));
  p->ctx = ctx; p->allocs = allocs; p->last_slot = -1; p->last_stream = ctx->stream;
  if (hipSetDevice(ctx->device) != hipSuccess) { delete p; return set_error(KAS_E_HIP, "hipSetDevice failed"); }
  for (int i = 0; i < KAS_TIMER_SLOTS; ++i) {
    if (hipEventCreate(&p->ev_start[i]) != hipSuccess || hipEventCreate(&p->ev_stop[i]) != hipSuccess ||
        hipEventCreate(&p->ev_mid[i]) != hipSuccess) {
      kas_plan_destroy(p);
      return set_error(KAS_E_HIP, "hipEventCreate failed");
    }
  }
  // (the hand-back count comes back through a word of pinned host memory: no copy, no synchronisation; without it the launch
  // behind the slim kernel keeps its small grid)
  if (hipHostMalloc((void**)&p->h_handback, sizeof(int32_t), hipHostMallocMapped) == hipSuccess) {
    *p->h_handback = 0;
    if (hipHostGetDevicePointer((void**)&p->d_handback, p->h_handback, 0) != hipSuccess) p->d_handback = nullptr;
  } else {
    (void)hipGetLastError();
    p->h_handback = nullptr;
  }
  // (a short run of the LDS lane-order self-test again: the property is checked where and when the form is about to be used)
  if (ctx->lds_lane_order_ok) {
    KasShape pre;
    std::string err;
    if (kas_shape_batch(batch, &pre, &err, 0, 0) == KAS_E_OK && pre.relax_ok) kas_lds_order_selftest(ctx, 1024, 40);
  }
  const int rc = kas_plan_build(p, batch);
  if (rc != KAS_E_OK) { kas_plan_destroy(p); return rc; }
  *out_plan = p;
  return KAS_E_OK;
}

int kas_plan_create(kas_ctx* ctx, const kas_batch_desc* batch, kas_plan** out_plan) {
  if (!ctx || !batch || !out_plan) return set_error(KAS_E_INVALID_ARG, "NULL argument");
  return kas_plan_new(ctx, batch, nullptr, out_plan);
}

// 16-bit cells are node indices: the node table the kernels see gives node i the id i (`ids` holds it, `out` = *b with it).
// `ids` may be a table an earlier call filled (the context's): it is rewritten only where the scenarios' node ranges
// differ from the ones it was filled for (`stamp`: a hash of them) — a what-if caller's 1000 x 1000 table stays.
static int kas_ident_batch(const kas_batch_desc* b, std::vector<int32_t>* ids, kas_batch_desc* out, uint64_t* stamp = nullptr) {
  if (b->n_scenarios < 0 || b->node_pool_len < 0 || (b->n_scenarios > 0 && !b->scenarios))
    return set_error(KAS_E_INVALID_ARG, "negative size / scenarios == NULL");
  uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)b->node_pool_len;
  for (int32_t s = 0; s < b->n_scenarios; ++s) {
    const kas_scenario_desc& sd = b->scenarios[s];
    if (sd.n_nodes < 0 || sd.node_off < 0 || sd.node_off + sd.n_nodes > b->node_pool_len)
      return set_error(KAS_E_INVALID_ARG, "scenario " + std::to_string(s) + ": node table outside the node pool");
    // (a cell's bit 15 says "no holder" inside the kernels, mid_to_index: an index is below 32,768 — KAS_N_LIMIT, the
    // limit of every plan; 0xFFFF is the only cell value above it that means anything)
    if (sd.n_nodes > KAS_N_LIMIT)
      return set_error(KAS_E_UNSUPPORTED, "scenario " + std::to_string(s) + ": more than 32,767 brokers do not fit 16-bit cells");
    h = (h ^ (((uint64_t)(uint32_t)sd.node_off << 32) | (uint32_t)sd.n_nodes)) * 0x100000001b3ull;
    h ^= h >> 29;
  }
  h |= 1ull;
  if (!(stamp && *stamp == h && ids->size() == (size_t)b->node_pool_len)) {
    ids->assign((size_t)b->node_pool_len, 0);
    for (int32_t s = 0; s < b->n_scenarios; ++s) {
      const kas_scenario_desc& sd = b->scenarios[s];
      for (int32_t i = 0; i < sd.n_nodes; ++i) (*ids)[(size_t)(sd.node_off + i)] = i;
    }
    if (stamp) *stamp = h;
  }
  *out = *b;
  out->node_id = ids->data();
  return KAS_E_OK;
}

int kas_plan_create16(kas_ctx* ctx, const kas_batch_desc* batch, kas_plan** out_plan) {
  if (!ctx || !batch || !out_plan) return set_error(KAS_E_INVALID_ARG, "NULL argument");
  std::vector<int32_t> ids;
  kas_batch_desc ib;
  const int rc = kas_ident_batch(batch, &ids, &ib);
  if (rc != KAS_E_OK) return rc;
  return kas_plan_new(ctx, &ib, nullptr, out_plan, 1);
}

// the launch decisions of kas_solve_device, in one place
struct KasLaunchPlan {
  bool p4_order;                // first fit inside the order kernel's workgroup (kas_p4_order_kernel; then `relax`, and no kas_p4_kernel)
  bool relaxw;                  // relaxation form of P5 for lists 4 and 5 wide (then neither tickets nor wide)
  bool relax;                   // relaxation form of P5 (then neither tickets nor wide)
  bool tickets, pairing, wide;
  bool m32;                     // dword mid rows in this solve (KAS_FLAG_MID32)
  int tiles;                    // relaxation form: 0 = tiles of 64 rows, 1 = double tiles, 2 = quad tiles (dword mid rows only)
  int packed;
  unsigned fill_grid, fill_block, order_grid, order_block;
  size_t fill_lds, order_lds;
};
static KasLaunchPlan kas_launch_plan(const kas_plan* p) {
  KasLaunchPlan lp;
  lp.relax = p->shape.relax_ok && p->ctx->lds_lane_order_ok && kas_order_relax_for(p->Wc, 0, 0) != nullptr &&
             !(p->flags & KAS_FLAG_ROUND_ORDER) && !(kas_flags_want_tickets(p->flags) && p->tickets);
  lp.tickets = !lp.relax && p->tickets && !(p->flags & KAS_FLAG_ROUND_ORDER) && !p->cells16;   // (no ticket form with 16-bit cells: the round form)
  lp.packed = p->shape.packed_ok && !(p->flags & KAS_FLAG_WIDE_COUNTERS);
  lp.pairing = lp.tickets && p->G > 1 && p->n_scenarios > p->G;
  lp.relaxw = !lp.relax && !p->cells16 && p->shape.relaxw_ok && p->ctx->lds_lane_order_ok && kas_order_relaxw_for(p->Wc) != nullptr &&
              kas_relaxw_wanted(p->flags);
  lp.wide = !lp.tickets && !lp.relaxw && !p->cells16 && p->shape.wide_ok && !(p->flags & KAS_FLAG_ROUND_ORDER) && kas_order_wide_for(p->Wc) != nullptr;
  lp.p4_order = false;
  lp.m32 = kas_mid32_launch(p->shape, p->cells16 != 0, p->mid32_bits, lp.relax, p->flags, kas_plan_relax_idl(p), kas_plan_index_rows(p),
                            kas_plan_spread_chunks(p)) && kas_order_relax_m32_pick(p->Wc, 0) != nullptr;
  lp.tiles = 0;
  lp.fill_grid = (unsigned)p->n_scenarios; lp.fill_block = 64u * (unsigned)p->NW;
  lp.fill_lds = (size_t)(kas_plan_fused(p) ? p->lds_fused.total : p->lds.total) + KAS_TUNE_FILL_LDS_PAD;
  if (lp.relax) {
    int dual = p->Wc == 3 && kas_relax_double_tiles(p->flags, p->n_scenarios);
    if (dual && lp.m32 && kas_relax_quad_tiles(p->flags, p->n_scenarios) && kas_order_relax_lds(p->shape.n_max, 2, 0, 1) <= KAS_LDS_LIMIT)
      dual = 2;                                                 // (quad tiles: the instances on dword mid rows)
    lp.tiles = dual;
    lp.order_grid = (unsigned)p->n_scenarios; lp.order_block = 64u;
    lp.order_lds = (size_t)kas_order_relax_lds(p->shape.n_max, dual, p->shape.any_ctx, kas_plan_relax_idl(p));
    // first fit as a second wavefront of the order kernel's workgroup?
    const bool relax_plain = !p->shape.any_ctx && (p->flags >> 24) == 0u && (p->cells16 || kas_plan_relax_idl(p)) &&
                             kas_p4_order_for(p->Wc, dual != 0, p->cells16) != nullptr && p->b_p4s.p != nullptr;
    if (kas_p4_with_order(p->shape, p->NW, p->flags | (p->shape.with_x ? 0u : KAS_FLAG_GENERIC_FILL), kas_plan_spread_chunks(p), p->n_scenarios,
                          relax_plain, dual, kas_plan_relax_idl(p))) {
      lp.p4_order = true;
      lp.order_block = 128u;
      lp.order_lds = (size_t)kas_p4_order_lds(p->shape.n_max, dual, kas_plan_relax_idl(p));
    }
  } else if (lp.relaxw) {
    lp.order_grid = (unsigned)p->n_scenarios; lp.order_block = 64u;
    lp.order_lds = (size_t)kas_order_relaxw_lds(p->shape.n_max, p->Wc);
  } else if (lp.tickets) {
    lp.order_grid = (unsigned)((p->n_scenarios + p->G - 1) / p->G); lp.order_block = 192u;
    lp.order_lds = (size_t)kas_order_ticket_lds(p->shape.n_max, p->G, lp.packed) + KAS_TUNE_ORDER_LDS_PAD;
  } else if (lp.wide) {
    lp.order_grid = (unsigned)p->n_scenarios; lp.order_block = (unsigned)KAS_ORDER_WIDE_BLOCK;
    lp.order_lds = (size_t)kas_order_wide_lds(p->shape.n_max);
  } else {
    lp.order_grid = (unsigned)p->n_scenarios; lp.order_block = 64u;
    lp.order_lds = (size_t)kas_order_round_lds(p->shape.n_max, p->Wc);
  }
  return lp;
}

// first fit in kas_p4_kernel in this plan's next solve (kas_split_p4)
static bool kas_plan_split_p4(const kas_plan* p) {
  return p->b_p4s.p != nullptr &&
         kas_split_p4(p->shape, p->NW, p->flags | (p->shape.with_x ? 0u : KAS_FLAG_GENERIC_FILL), kas_plan_spread_chunks(p), p->n_scenarios);
}

// dword mid rows in this plan's next solve (KAS_FLAG_MID32, kas_mid32_launch)
static bool kas_plan_mid32(const kas_plan*, const KasLaunchPlan& lp) { return lp.m32; }

int kas_plan_describe(const kas_plan* p, char* buf, int n) {
  if (!p || !buf || n <= 0) return set_error(KAS_E_INVALID_ARG, "NULL argument");
  const KasLaunchPlan lp = kas_launch_plan(p);
  const bool generic = (p->flags & KAS_FLAG_GENERIC_FILL) || !p->shape.with_x;
  char order[256];
  const char* ctx_tail = (lp.wide && p->shape.wide_checked)
                             ? " [count fields checked at the end; kas_fill_kernel + kas_order_round_kernel for scenarios it flags]"
                             : (p->shape.any_ctx && (lp.tickets || lp.wide || lp.relax)) ? " [Context in/out; kas_order_round_kernel for scenarios it flags]" : "";
  if (lp.relax && lp.p4_order)
    snprintf(order, sizeof(order), "kas_p4_order_kernel<%d>[first fit beside kas_order_relax_kernel<%d>[tiles of %d rows%s] in one workgroup] grid=%ux%u lds=%zu",
             p->Wc, p->Wc, 64 << lp.tiles,
             kas_plan_relax_idl(p) ? (kas_plan_mid32(p, lp) ? ", ids in LDS, dword mid rows" : ", ids in LDS") : "",
             lp.order_grid, lp.order_block, lp.order_lds);
  else if (lp.relax)
    snprintf(order, sizeof(order), "kas_order_relax_kernel<%d>[tiles of %d rows%s%s] grid=%ux%u lds=%zu%s", p->Wc,
             64 << lp.tiles,
             kas_plan_relax_idl(p) ? (kas_plan_mid32(p, lp) ? ", ids in LDS, dword mid rows" : ", ids in LDS") : "",
             (p->flags >> 24) ? ", sampled verification" : "", lp.order_grid, lp.order_block, lp.order_lds, ctx_tail);
  else if (lp.relaxw)
    snprintf(order, sizeof(order), "kas_order_relax_wide_kernel<%d>[tiles of 64 rows, ids in LDS] grid=%ux%u lds=%zu", p->Wc, lp.order_grid,
             lp.order_block, lp.order_lds);
  else if (lp.tickets)
    snprintf(order, sizeof(order), "%skas_order_ticket_kernel<%d,%d,%s> grid=%ux%u lds=%zu%s",
             lp.pairing ? "kas_order_permutation_kernel + " : "", p->Wc, p->G, lp.packed ? "true" : "false",
             lp.order_grid, lp.order_block, lp.order_lds, ctx_tail);
  else if (lp.wide)
    snprintf(order, sizeof(order), "kas_order_wide_kernel<%d> grid=%ux%u lds=%zu%s", p->Wc, lp.order_grid,
             lp.order_block, lp.order_lds, ctx_tail);
  else
    snprintf(order, sizeof(order), "kas_order_round_kernel<%d> grid=%ux%u lds=%zu", p->Wc, lp.order_grid,
             lp.order_block, lp.order_lds);
  const int32_t chunks = kas_plan_spread_chunks(p);
  char spread[128];
  spread[0] = 0;
  if (chunks > 0)
    snprintf(spread, sizeof(spread), "kas_spread_{a,q,b,p4}_kernel<%d> %d chunks x %d scenarios (rows not rack-diverse: ", p->Wc,
             chunks, p->n_scenarios);
  if (p->shape.relax_ok && !lp.relax && !(p->flags & KAS_FLAG_ROUND_ORDER) && !kas_flags_want_tickets(p->flags)) {
    const int stt = p->ctx->lds_lane_order_state;              // the form the shape allows is not the one launched: say why
    const size_t ol = strlen(order);
    snprintf(order + ol, sizeof(order) - ol, " [relaxation form off: LDS lane-order self-test %s]",
             stt == 0 ? "FAILED" : (stt == -2 ? "switched off (KAS_NO_LANE_ORDER)" : "could not run"));
  }
  char p4[96];
  p4[0] = 0;
  if (kas_plan_split_p4(p) && !lp.p4_order)                    // first fit (P4) is a launch of its own between the two
    snprintf(p4, sizeof(p4), " + kas_p4_kernel<%d> grid=%ux%u lds=%zu", p->Wc, (unsigned)p->n_scenarios, 64u * KAS_P4_KERNEL_WAVES,
             (size_t)kas_p4_lds_layout(p->shape.n_max).total);
  if (kas_plan_slim_fill(p, kas_plan_split_p4(p) || lp.p4_order, chunks)) {
    const unsigned back = kas_plan_back_grid(p, lp.fill_grid);
    const int len = snprintf(buf, (size_t)n, "kas_fill_slim_kernel<%d>[quota, chunk histograms] grid=%ux%u lds=%zu (+ kas_fill_kernel<%d,%d>[quota, chunk histograms] "
                             "grid=%ux%u lds=%zu for scenarios it hands back)%s + %s", p->Wc, lp.fill_grid, lp.fill_block,
                             (size_t)kas_fill_slim_lds(p->shape.n_max, p->Wc, p->shape.idmap_entries).total, p->Wc, p->NW, back,
                             lp.fill_block, lp.fill_lds, p4, order);
    return len < n ? len : n - 1;
  }
  const int len = snprintf(buf, (size_t)n, "%skas_fill_kernel<%d,%d>[%s] grid=%ux%u lds=%zu%s%s + %s%s", spread, p->Wc, p->NW,
                           generic ? "sweeps" : (kas_plan_fused(p) ? (kas_plan_index_rows(p) && chunks == 0 ? "quota, chunk histograms, index rows" : "quota, chunk histograms") : "quota"), lp.fill_grid,
                           lp.fill_block, lp.fill_lds, chunks > 0 ? ")" : "", p4, order, p->cells16 ? " [16-bit cells]" : "");
  return len < n ? len : n - 1;
}

int64_t kas_plan_algorithmic_bytes(const kas_plan* plan) {
  if (!plan) return -1;
  return plan->cells16 ? plan->shape.algorithmic_bytes16 : plan->shape.algorithmic_bytes;   // (the plan's own cell width)
}

static int kas_solve_device_impl(kas_plan* p, const kas_tables* t, void* hip_stream);

int kas_solve_device(kas_plan* p, const kas_tables* t, void* hip_stream) {
  if (!p || !t) return set_error(KAS_E_INVALID_ARG, "NULL argument");
  if (p->cells16) return set_error(KAS_E_INVALID_ARG, "a plan of kas_plan_create16 is solved by kas_solve_device16");
  return kas_solve_device_impl(p, t, hip_stream);
}

int kas_solve_device16(kas_plan* p, const kas_tables16* t16, void* hip_stream) {
  if (!p || !t16) return set_error(KAS_E_INVALID_ARG, "NULL argument");
  if (!p->cells16) return set_error(KAS_E_INVALID_ARG, "kas_solve_device16 needs a plan of kas_plan_create16");
  kas_tables t;                                              // (the kernels take the cell width from KAS_FLAG_CELLS16)
  memset(&t, 0, sizeof(t));
  t.cur = reinterpret_cast<const int32_t*>(t16->cur); t.out = reinterpret_cast<int32_t*>(t16->out);
  t.aux = t16->aux; t.ctx = t16->ctx; t.topic_results = t16->topic_results; t.scenario_results = t16->scenario_results;
  t.cur_len = t16->cur_len; t.out_len = t16->out_len; t.aux_len = t16->aux_len; t.ctx_len = t16->ctx_len;
  return kas_solve_device_impl(p, &t, hip_stream);
}

static int kas_solve_device_impl(kas_plan* p, const kas_tables* t, void* hip_stream) {
  if (p->n_scenarios == 0) return KAS_E_OK;
  if (!t->out || !t->topic_results || !t->scenario_results || (p->shape.cur_need > 0 && !t->cur) ||
      (p->shape.aux_need > 0 && !t->aux) || (p->shape.ctx_need > 0 && !t->ctx))
    return set_error(KAS_E_INVALID_ARG, "a table the descriptors refer to is NULL");
  KAS_HIP_TRY(hipSetDevice(p->ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : p->ctx->stream;
  KasLaunch a;
  a.scen = (const kas_scenario_desc*)p->b_scen.p; a.topics = (const kas_topic_desc*)p->b_topics.p;
  a.node_id = (const int32_t*)p->b_node_id.p; a.node_rack = (const int32_t*)p->b_node_rack.p;
  a.cur = t->cur; a.out = t->out; a.aux = t->aux; a.ctx = t->ctx;
  a.topic_results = t->topic_results; a.scenario_results = t->scenario_results;
  a.accmask = (uint64_t*)p->b_accmask.p; a.accmask_off = (const int64_t*)p->b_accmask_off.p;
  a.stats = (int64_t*)p->b_stats.p;
  a.orph = (int32_t*)p->b_orph.p; a.orph_off = (const int64_t*)p->b_orph_off.p;
  a.perm = nullptr;
  a.ord_flag = (int32_t*)p->b_ord_flag.p;
  // the plan's scratch serves one solve at a time: order this solve behind the previous one
  if (p->last_slot >= 0 && p->last_stream != st)
    KAS_HIP_TRY(hipStreamWaitEvent(st, p->ev_stop[p->last_slot], 0));
  p->last_stream = st;
  a.n_scenarios = p->n_scenarios; a.n_max = p->shape.n_max;
  a.idmap_entries = p->shape.idmap_entries; a.need_bsearch = p->shape.need_bsearch;
  a.flags = (p->flags & ~(KAS_FLAG_FUSED_HIST | KAS_FLAG_ORDER_FLAGGED | KAS_FLAG_WIDE_CHECK)) | (p->shape.with_x ? 0u : KAS_FLAG_GENERIC_FILL) |
            (kas_plan_fused(p) ? KAS_FLAG_FUSED_HIST : 0u) |
            (kas_relax_double_tiles(p->flags, p->n_scenarios) ? KAS_FLAG_RELAX_DUAL : 0u) |
            ((p->ctx->lds_lane_order_ok && !(p->flags & KAS_FLAG_NO_RTN_QUOTA)) ? KAS_FLAG_LANE_ORDER : 0u) |
            (p->cells16 ? KAS_FLAG_CELLS16 : 0u) | (kas_plan_index_rows(p) ? KAS_FLAG_INDEX_ROWS : 0u);
  const KasLaunchPlan lp = kas_launch_plan(p);
  const bool m32 = kas_plan_mid32(p, lp);                      // every kernel of this solve moves mid rows as one dword each
  if (m32) a.flags |= KAS_FLAG_MID32;
  const bool tickets = lp.tickets;
  const bool split_p4 = kas_plan_split_p4(p) || lp.p4_order;   // (the fill kernel hands first fit over: to kas_p4_kernel, or to kas_p4_order_kernel)
  a.p4s = (int32_t*)p->b_p4s.p;
  a.flags = split_p4 ? (a.flags | KAS_FLAG_SPLIT_P4) : (a.flags & ~KAS_FLAG_SPLIT_P4);   // (the kernels' bit: this launch's form)
  const int slot = p->timer_next;
  a.sp_hist = nullptr; a.sp_quota = nullptr; a.sp_node = nullptr; a.sp_flag = nullptr; a.sp_oc = nullptr; a.sp_chunks = 0;
  a.handback = nullptr;
  const int32_t chunks = kas_plan_spread_chunks(p);
  if (chunks != p->sp_alloc_chunks)
    return set_error(KAS_E_INVALID_ARG, "internal: spread-fill scratch not sized for this plan state");
  KAS_HIP_TRY(hipEventRecord(p->ev_start[slot], st));
  if (chunks > 0) {
    const KasSpreadKernels sk = kas_spread_for(p->Wc);
    a.sp_hist = (int32_t*)p->b_sp_hist.p; a.sp_quota = (int32_t*)p->b_sp_quota.p; a.sp_node = (int32_t*)p->b_sp_node.p;
    a.sp_flag = (int32_t*)p->b_sp_flag.p; a.sp_oc = (int32_t*)p->b_sp_oc.p; a.sp_chunks = chunks;
    KAS_HIP_TRY(hipMemsetAsync(a.sp_flag, 0, 4 * ((size_t)p->n_scenarios + 1), st));
    KAS_HIP_TRY(hipMemsetAsync(a.sp_oc, 0, 4 * (size_t)p->n_scenarios * ((size_t)chunks + 2), st));
    const size_t la = (size_t)kas_spread_scan_lds(p->shape.n_max, p->Wc, p->shape.idmap_entries, p->shape.need_bsearch, 1).total;
    const size_t lb = (size_t)kas_spread_scan_lds(p->shape.n_max, p->Wc, p->shape.idmap_entries, p->shape.need_bsearch, 2).total;
    const size_t l4 = (size_t)kas_spread_scan_lds(p->shape.n_max, p->Wc, p->shape.idmap_entries, p->shape.need_bsearch, 3).total;
    const dim3 gc((unsigned)chunks, (unsigned)p->n_scenarios);
    hipLaunchKernelGGL(sk.a, gc, dim3(64), la, st, a);
    hipLaunchKernelGGL(sk.q, dim3((unsigned)((p->shape.n_max + 255) / 256), (unsigned)p->n_scenarios), dim3(256), 0, st, a);
    hipLaunchKernelGGL(sk.b, gc, dim3(64), lb, st, a);
    hipLaunchKernelGGL(sk.p4, dim3((unsigned)p->n_scenarios), dim3(64 * KAS_SPREAD_P4_WAVES), l4, st, a);
    KAS_HIP_TRY(hipGetLastError());
    a.flags |= KAS_FLAG_ONLY_FLAGGED;                              // what is left: scenarios handed back (not rack-diverse, ...)
  }
  // (tuning builds, scripts/build_variant.sh -- -DKAS_TUNE_...: one of the two kernels alone, to see what each saturates at.
  // ORDER_ONLY: the fill runs in the plan's first solve only, and KAS_TUNE_NO_ROW_STORES keeps its mid rows in place.)
  unsigned fill_grid = lp.fill_grid;
  if (kas_plan_slim_fill(p, split_p4, chunks)) {
    // the slim kernel takes every scenario (and writes each one's hand-back flag, 0 or 1); the full kernel behind it takes the
    // flagged ones on a small grid — its workgroups need a slot of 35 KB of LDS each before they can see that there is nothing to do
    a.sp_flag = (int32_t*)p->b_sp_flag.p;
#if defined(KAS_TUNE_ORDER_ONLY)
    if (p->last_slot < 0)
#endif
    hipLaunchKernelGGL(m32 ? kas_fill_slim_m32() : kas_fill_slim_for(p->Wc), dim3(lp.fill_grid), dim3(lp.fill_block),
                       (size_t)kas_fill_slim_lds(p->shape.n_max, p->Wc, p->shape.idmap_entries).total + KAS_TUNE_SLIM_LDS_PAD, st, a);
    KAS_HIP_TRY(hipGetLastError());
    a.flags |= KAS_FLAG_ONLY_FLAGGED;
    fill_grid = kas_plan_back_grid(p, fill_grid);
    a.handback = p->d_handback;
  }
#if defined(KAS_TUNE_ORDER_ONLY)
  if (p->last_slot < 0)
#endif
  hipLaunchKernelGGL(kas_fill_for(p->Wc, p->NW), dim3(fill_grid), dim3(lp.fill_block), lp.fill_lds, st, a);
  KAS_HIP_TRY(hipGetLastError());
  a.flags &= ~KAS_FLAG_ONLY_FLAGGED;
  if (split_p4 && !lp.p4_order) {
    hipLaunchKernelGGL(m32 ? kas_p4_m32() : kas_p4_for(p->Wc), dim3((unsigned)p->n_scenarios), dim3(64 * KAS_P4_KERNEL_WAVES),
                       (size_t)kas_p4_lds_layout(p->shape.n_max).total, st, a);
    KAS_HIP_TRY(hipGetLastError());
  }
  KAS_HIP_TRY(hipEventRecord(p->ev_mid[slot], st));
  const int packed = lp.packed;
  // a Context handed in: the ticket forms flag the scenarios whose counters do not fit their count
  // fields, and the round form (launched behind them, taking only those) serves them
  // lists 4-5 wide and a node that may hold 1023 .. 2039 rows: the wide form checks its 10-bit count fields when
  // the last row has retired and flags a scenario that outgrew them.  Its rows are finished (on wrong counts) by
  // then and its mid rows gone, so it is solved again from `cur`: fill kernel, then round form, both taking only
  // the flagged scenarios.
  const bool wide_recheck = lp.wide && p->shape.wide_checked;
  const bool ctx_fallback = (p->shape.any_ctx && (tickets || lp.wide || lp.relax)) || wide_recheck;
  if (ctx_fallback) KAS_HIP_TRY(hipMemsetAsync(a.ord_flag, 0, 4 * ((size_t)p->n_scenarios + 1), st));
  if (wide_recheck) a.flags |= KAS_FLAG_WIDE_CHECK;
  if (lp.pairing) {
    // scenarios that share a solver wavefront should have P5 chains of similar length
    a.perm = (int32_t*)p->b_perm.p;
    hipLaunchKernelGGL(kas_order_permutation_kernel, dim3(1), dim3(64 * KAS_PERM_WAVES),
                       sizeof(int32_t) * (size_t)(KAS_PERM_BINS + 8), st, a);
    KAS_HIP_TRY(hipGetLastError());
  }
#if defined(KAS_TUNE_SKIP_ORDER)
  if (true) {}
  else
#endif
  if (lp.relax && lp.p4_order)
    hipLaunchKernelGGL(m32 ? kas_p4_order_m32_pick(p->Wc, lp.tiles) : kas_p4_order_for(p->Wc, (a.flags & KAS_FLAG_RELAX_DUAL) != 0u, p->cells16),
                       dim3(lp.order_grid), dim3(lp.order_block), lp.order_lds, st, a);
  else if (lp.relax && m32)
    hipLaunchKernelGGL(kas_order_relax_m32_pick(p->Wc, lp.tiles), dim3(lp.order_grid), dim3(lp.order_block), lp.order_lds, st, a);
  else if (lp.relax)
    hipLaunchKernelGGL(kas_order_relax_for(p->Wc, (a.flags & KAS_FLAG_RELAX_DUAL) != 0u, p->shape.any_ctx, (a.flags >> 24) != 0u, p->cells16, kas_plan_relax_idl(p)),
                       dim3(lp.order_grid), dim3(lp.order_block), lp.order_lds, st, a);
  else if (lp.relaxw)
    hipLaunchKernelGGL(kas_order_relaxw_for(p->Wc), dim3(lp.order_grid), dim3(lp.order_block), lp.order_lds, st, a);
  else if (tickets)
    hipLaunchKernelGGL(kas_order_ticket_for(p->Wc, p->G, packed), dim3(lp.order_grid), dim3(lp.order_block),
                       lp.order_lds, st, a);
  else if (lp.wide)
    hipLaunchKernelGGL(kas_order_wide_for(p->Wc), dim3(lp.order_grid), dim3(lp.order_block), lp.order_lds, st, a);
  else
    hipLaunchKernelGGL(kas_order_round_for(p->Wc), dim3(lp.order_grid), dim3(lp.order_block), lp.order_lds, st, a);
  KAS_HIP_TRY(hipGetLastError());
  if (wide_recheck) {
    KasLaunch af = a;
    af.flags = (af.flags | KAS_FLAG_ONLY_FLAGGED) & ~(KAS_FLAG_WIDE_CHECK | KAS_FLAG_SPLIT_P4);   // (this fill does its own first fit)
    af.sp_flag = a.ord_flag;                                 // (same meaning: != 0, this kernel takes the scenario)
    af.handback = nullptr;
    hipLaunchKernelGGL(kas_fill_for(p->Wc, p->NW), dim3(lp.fill_grid), dim3(lp.fill_block), lp.fill_lds, st, af);
    KAS_HIP_TRY(hipGetLastError());
  }
  if (ctx_fallback) {
    a.flags |= KAS_FLAG_ORDER_FLAGGED;
    a.perm = nullptr;
    hipLaunchKernelGGL(kas_order_round_for(p->Wc), dim3((unsigned)p->n_scenarios), dim3(64),
                       (size_t)kas_order_round_lds(p->shape.n_max, p->Wc), st, a);
    KAS_HIP_TRY(hipGetLastError());
  }
  KAS_HIP_TRY(hipEventRecord(p->ev_stop[slot], st));
  p->last_slot = slot;
  p->timer_next = (slot + 1) % KAS_TIMER_SLOTS;
  if (p->timer_count < KAS_TIMER_SLOTS) p->timer_count += 1;
  return KAS_E_OK;
}

static int kas_plan_times(kas_plan* p, double* fill_us, double* order_us, int* launches) {
  *fill_us = 0.0; *order_us = 0.0; *launches = 0;
  KAS_HIP_TRY(hipSetDevice(p->ctx->device));
  double f_ms = 0.0, o_ms = 0.0;
  int n = 0;
  for (int i = 0; i < p->timer_count; ++i) {
    int slot = (p->timer_next - 1 - i + 2 * KAS_TIMER_SLOTS) % KAS_TIMER_SLOTS;
    KAS_HIP_TRY(hipEventSynchronize(p->ev_stop[slot]));
    float ms = 0.f;
    KAS_HIP_TRY(hipEventElapsedTime(&ms, p->ev_start[slot], p->ev_mid[slot]));
    f_ms += ms;
    KAS_HIP_TRY(hipEventElapsedTime(&ms, p->ev_mid[slot], p->ev_stop[slot]));
    o_ms += ms;
    ++n;
  }
  p->timer_count = 0;
  *launches = n;
  if (n) { *fill_us = f_ms * 1000.0 / n; *order_us = o_ms * 1000.0 / n; }
  return KAS_E_OK;
}

int kas_plan_kernel_time_us(kas_plan* p, double* avg_us, int* launches) {
  if (!p || !avg_us || !launches) return set_error(KAS_E_INVALID_ARG, "NULL argument");
  double f = 0.0, o = 0.0;
  int rc = kas_plan_times(p, &f, &o, launches);
  *avg_us = f + o;
  return rc;
}

int kas_plan_phase_times_us(kas_plan* p, double* fill_us, double* order_us, int* launches) {
  if (!p || !fill_us || !order_us || !launches) return set_error(KAS_E_INVALID_ARG, "NULL argument");
  return kas_plan_times(p, fill_us, order_us, launches);
}

int kas_plan_set_flags(kas_plan* p, uint32_t flags) {
  if (!p) return set_error(KAS_E_INVALID_ARG, "plan == NULL");
  const int nw = (int)((flags >> 8) & 0xfu), g = (int)((flags >> 12) & 0xfu);
  if (nw != 0 && nw != 1 && nw != 2 && nw != 4)
    return set_error(KAS_E_INVALID_ARG, "KAS_PLAN_WAVES: waves per scenario must be 1, 2 or 4");
  if (g != 0 && g != 1 && g != 2 && g != 4)
    return set_error(KAS_E_INVALID_ARG, "KAS_PLAN_GROUPS: scenarios per wavefront must be 1, 2 or 4");
  if ((flags & KAS_FLAG_ROUND_ORDER) && !p->shape.round_fits)
    return set_error(KAS_E_UNSUPPORTED, "KAS_PLAN_ROUND_ORDER: the round form's LDS exceeds 160 KiB at this broker count x width");
  if ((flags >> 24) != 0u && p->shape.relax_ok && kas_order_relax_for(p->Wc, 0, p->shape.any_ctx, 1, p->cells16, kas_plan_relax_idl(p)) == nullptr)
    return set_error(KAS_E_UNSUPPORTED, "KAS_PLAN_VERIFY_SAMPLE: not instantiated for the instances that gather the broker ids from the node table (this many brokers)");
  if (p->cells16 && kas_flags_want_tickets(flags) && !p->shape.round_fits)
    return set_error(KAS_E_UNSUPPORTED, "16-bit cells: no ticket form; the round form it would take does not fit at this broker count");
  if (!kas_minimal_ok(p->Wc, nw ? nw : p->NW, g ? g : p->G))
    return set_error(KAS_E_UNSUPPORTED, "tuning build (KAS_MINIMAL_INSTANCES): only 4 fill waves, 1 or 2 groups");
  const KasShape& sh = p->shape;
  // (every refusal before the plan is touched: a call that returns an error leaves the plan as it was — ADVICE r4)
  KasLds l_nw = p->lds;
  if (nw != 0 && nw != p->NW) {
    l_nw = kas_fill_lds_layout(sh.n_max, sh.Wc, nw, sh.idmap_entries, sh.need_bsearch, sh.with_x);
    if (l_nw.total > KAS_LDS_LIMIT)
      return set_error(KAS_E_UNSUPPORTED, "LDS carve-up exceeds 160 KiB at that many waves");
  }
  if (g != 0 && g != p->G &&
      (kas_order_ticket_lds(sh.n_max, g, 0) > KAS_LDS_LIMIT || (int64_t)g * kas_order_ticket_group_bytes(sh.n_max, g, 0) > 65536))
    return set_error(KAS_E_UNSUPPORTED, "LDS of the order kernel exceeds 160 KiB at that many groups");
  if (nw != 0 && nw != p->NW) {
    p->lds = l_nw;
    p->NW = nw;
    KasShape tmp = sh;                                       // per-chunk histograms at the new workgroup width?
    tmp.NW = nw; tmp.lds = l_nw;
    kas_choose_fused(&tmp);
    p->fused = tmp.fused_ok; p->lds_fused = tmp.lds_fused;
  }
  if (g != 0 && g != p->G) p->G = g;
  KAS_HIP_TRY(hipSetDevice(p->ctx->device));
  int rc = kas_plan_set_kernels(p);
  if (rc != KAS_E_OK) return rc;
  p->index_rows_bits = flags & (KAS_PLAN_NO_INDEX_ROWS_BIT | KAS_PLAN_INDEX_ROWS_BIT);
  p->mid32_bits = flags & (KAS_PLAN_NO_MID32_BIT | KAS_PLAN_MID32_BIT);
  p->full_fill = flags & KAS_PLAN_FULL_FILL_BIT;
  p->flags = (flags & (0xff0000ffu | KAS_FLAG_TICKET_ORDER | KAS_FLAG_RELAX_TILES_64 | KAS_FLAG_RELAX_TILES_128 | KAS_FLAG_NO_RTN_QUOTA | KAS_FLAG_FILL_WITH_P4 | KAS_FLAG_SPLIT_P4) & ~(KAS_FLAG_FUSED_HIST | KAS_FLAG_ONLY_FLAGGED | KAS_FLAG_ORDER_FLAGGED)) |
             (g != 0 ? KAS_FLAG_TICKET_ORDER : 0u);      // (scenarios per wavefront only mean something to the ticket form)
  // the spread fill's scratch follows the flags (allocated here, never inside a solve); a solve of this
  // plan may still be in flight on the old scratch
  if (kas_plan_spread_chunks(p) != p->sp_alloc_chunks) {
    if (p->last_slot >= 0) KAS_HIP_TRY(hipEventSynchronize(p->ev_stop[p->last_slot]));
    if ((rc = kas_plan_spread_scratch(p)) != KAS_E_OK) return rc;
  }
  return KAS_E_OK;
}

int kas_plan_stats(kas_plan* p, int64_t* out, int64_t n) {
  if (!p || !out) return set_error(KAS_E_INVALID_ARG, "NULL argument");
  const int64_t need = (int64_t)KAS_STATS_PER_SCENARIO * p->n_scenarios;
  if (n < need) return set_error(KAS_E_INVALID_ARG, "stats buffer too small");
  KAS_HIP_TRY(hipSetDevice(p->ctx->device));
  KAS_HIP_TRY(hipStreamSynchronize(p->last_stream));
  if (need > 0) KAS_HIP_TRY(hipMemcpy(out, p->b_stats.p, sizeof(int64_t) * (size_t)need, hipMemcpyDeviceToHost));
  return KAS_E_OK;
}

// ---------------------------------------------------------------------------------------------
// host path
// ---------------------------------------------------------------------------------------------
int kas_host_alloc(int64_t bytes, void** out_ptr) {
  if (!out_ptr || bytes < 0) return set_error(KAS_E_INVALID_ARG, "kas_host_alloc: NULL / negative size");
  *out_ptr = nullptr;
  hipError_t e = hipHostMalloc(out_ptr, (size_t)(bytes > 0 ? bytes : 16), hipHostMallocPortable);
  if (e != hipSuccess) { *out_ptr = nullptr; return set_error(KAS_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
  return KAS_E_OK;
}

void kas_host_free(void* ptr) { if (ptr) (void)hipHostFree(ptr); }

void kas_shard_range(int64_t total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi) {
  if (world < 1) world = 1;
  if (rank < 0) rank = 0;
  if (rank >= world) rank = world - 1;
  if (total < 0) total = 0;
  const int64_t base = total / world, rem = total % world;
  const int64_t l = (int64_t)rank * base + (rank < rem ? rank : rem);
  if (lo) *lo = l;
  if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

int kas_batch_slice(const kas_batch_desc* b, int64_t lo, int64_t hi, kas_scenario_desc* scratch, kas_batch_desc* out,
                    const kas_tables* tables, kas_tables* tables_out) {
  if (!b || !out || lo < 0 || hi < lo || hi > b->n_scenarios || (hi > lo && (!scratch || !b->scenarios)))
    return set_error(KAS_E_INVALID_ARG, "kas_batch_slice: bad range / NULL argument");
  int64_t tlo = INT64_MAX, thi = 0, nlo = INT64_MAX, nhi = 0;
  for (int64_t i = lo; i < hi; ++i) {
    const kas_scenario_desc& sd = b->scenarios[i];
    if (sd.topic_begin < 0 || sd.topic_count < 0 || (int64_t)sd.topic_begin + sd.topic_count > b->n_topics || sd.n_nodes < 0 ||
        sd.node_off < 0 || sd.node_off + sd.n_nodes > b->node_pool_len)
      return set_error(KAS_E_INVALID_ARG, "kas_batch_slice: scenario " + std::to_string(i) + " refers outside the batch");
    if (sd.topic_count > 0) {
      if (sd.topic_begin < tlo) tlo = sd.topic_begin;
      if ((int64_t)sd.topic_begin + sd.topic_count > thi) thi = (int64_t)sd.topic_begin + sd.topic_count;
    }
    if (sd.n_nodes > 0) {
      if (sd.node_off < nlo) nlo = sd.node_off;
      if (sd.node_off + sd.n_nodes > nhi) nhi = sd.node_off + sd.n_nodes;
    }
  }
  if (tlo > thi) tlo = thi = 0;
  if (nlo > nhi) nlo = nhi = 0;
  for (int64_t i = lo; i < hi; ++i) {
    kas_scenario_desc sd = b->scenarios[i];
    sd.topic_begin = sd.topic_count > 0 ? (int32_t)(sd.topic_begin - tlo) : 0;
    sd.node_off = sd.n_nodes > 0 ? sd.node_off - nlo : 0;
    scratch[i - lo] = sd;
  }
  out->n_scenarios = (int32_t)(hi - lo);
  out->n_topics = (int32_t)(thi - tlo);
  out->scenarios = scratch;
  out->topics = b->topics ? b->topics + tlo : nullptr;
  out->node_id = b->node_id ? b->node_id + nlo : nullptr;
  out->node_rack = b->node_rack ? b->node_rack + nlo : nullptr;
  out->node_pool_len = nhi - nlo;
  if (tables && tables_out) {
    *tables_out = *tables;
    tables_out->topic_results = tables->topic_results ? tables->topic_results + tlo : nullptr;
    tables_out->scenario_results = tables->scenario_results ? tables->scenario_results + lo : nullptr;
  }
  return KAS_E_OK;
}

// word-at-a-time hash of the descriptor bytes (a what-if call hashes megabytes of node tables)
// (four independent lanes over 32-byte blocks: one dependent multiply chain ran at ~4 GB/s and made the 8 MB of
// descriptors and node tables of a 1000-variant what-if call cost 2 of its 5 ms)
static uint64_t kas_hash64(uint64_t h, const void* data, size_t n) {
  const unsigned char* p = (const unsigned char*)data;
  const uint64_t M = 0x9E3779B97F4A7C15ull;
  uint64_t a = h, b = h ^ 0x243F6A8885A308D3ull, c = h ^ 0x13198A2E03707344ull, d = h ^ 0xA4093822299F31D0ull;
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    uint64_t w[4];
    memcpy(w, p + i, 32);
    a = (a ^ w[0]) * M; a ^= a >> 29;
    b = (b ^ w[1]) * M; b ^= b >> 29;
    c = (c ^ w[2]) * M; c ^= c >> 29;
    d = (d ^ w[3]) * M; d ^= d >> 29;
  }
  h = a;
  h = (h ^ b) * M; h ^= h >> 29;
  h = (h ^ c) * M; h ^= h >> 29;
  h = (h ^ d) * M; h ^= h >> 29;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, p + i, 8);
    h = (h ^ w) * M;
    h ^= h >> 29;
  }
  for (; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
  return h;
}

// The plan of this batch from the context's cache: the one whose descriptors and node tables are the
// same byte for byte, else the least recently used entry rebuilt in place (or a new one into a free entry).
static int kas_host_plan(kas_ctx* ctx, const kas_batch_desc* b, kas_plan** out_plan, int cells16 = 0) {
  *out_plan = nullptr;
  if (b->n_scenarios < 0 || b->n_topics < 0 || b->node_pool_len < 0 ||
      (b->n_scenarios > 0 && !b->scenarios) || (b->n_topics > 0 && !b->topics) ||
      (b->node_pool_len > 0 && (!b->node_id || !b->node_rack)))
    return set_error(KAS_E_INVALID_ARG, "null/negative batch");
  const size_t sb = sizeof(kas_scenario_desc) * (size_t)b->n_scenarios, tb = sizeof(kas_topic_desc) * (size_t)b->n_topics,
               nb = sizeof(int32_t) * (size_t)b->node_pool_len;
  // what identifies the batch: (S, T, node pool length), scenario and topic descriptors, node tables — hashed and
  // compared where they lie; a copy is made only when a plan is built for them
  int64_t hdr[2] = {((int64_t)b->n_scenarios << 32) | (uint32_t)b->n_topics, b->node_pool_len | ((int64_t)(cells16 ? 1 : 0) << 62)};   // (a plan for 16-bit cells is another plan)
  const void* seg[5] = {hdr, b->scenarios, b->topics, b->node_id, b->node_rack};
  const size_t seg_bytes[5] = {16, sb, tb, nb, nb};
  const size_t desc_bytes = 16 + sb + tb + 2 * nb;
  uint64_t key = 0xcbf29ce484222325ull;
  for (int i = 0; i < 5; ++i) if (seg_bytes[i]) key = kas_hash64(key, seg[i], seg_bytes[i]);
  // (S, T) + topic descriptors + the cell width: a plan for the other cell width is never the one rebuilt in place
  const uint64_t sig = kas_hash64(0x84222325cbf29ce4ull, hdr, 8) ^ kas_hash64(0, b->topics, tb) ^ (cells16 ? 0x5bd1e995c16c16c1ull : 0ull);
  auto same_bytes = [&](const std::vector<unsigned char>& have) {
    if (have.size() != desc_bytes) return false;
    size_t off = 0;
    for (int i = 0; i < 5; ++i) {
      if (seg_bytes[i] && memcmp(have.data() + off, seg[i], seg_bytes[i]) != 0) return false;
      off += seg_bytes[i];
    }
    return true;
  };
  ctx->use_clock += 1;
  // hit: the same bytes.  Miss: rebuild the least recently used plan of the same signature in place (a
  // what-if caller: same snapshot, other broker sets — every buffer is already large enough), else fill a
  // free entry, else rebuild the least recently used plan of any shape.  Entries this call already uses
  // (the other scenario ranges of a split call) are never victims.
  KasCachedPlan *same = nullptr, *empty = nullptr, *lru = nullptr;
  for (KasCachedPlan& c : ctx->plans) {
    if (c.plan && c.key == key && same_bytes(c.desc)) {
      c.last_use = ctx->use_clock; c.call = ctx->host_calls;
      ctx->host_plan_hits += 1;
      *out_plan = c.plan;
      return KAS_E_OK;                                         // (its flags are the ones set below when it was built)
    }
    if (!c.plan) { if (!empty) empty = &c; continue; }
    if (c.call == ctx->host_calls) continue;
    if (c.sig == sig && (!same || c.last_use < same->last_use)) same = &c;
    if (!lru || c.last_use < lru->last_use) lru = &c;
  }
  KasCachedPlan* victim = same ? same : (empty ? empty : lru);
  if (!victim) return set_error(KAS_E_NOMEM, "host-path plan cache exhausted by one call");
  int rc;
  if (victim->plan) {
    victim->plan->cells16 = cells16;
    rc = kas_plan_build(victim->plan, b);                      // in place: its buffers are reused where they are large enough
    if (rc != KAS_E_OK) { kas_plan_destroy(victim->plan); victim->plan = nullptr; victim->desc.clear(); victim->sig = 0; return rc; }
  } else {
    kas_plan* plan = nullptr;
    rc = kas_plan_new(ctx, b, &ctx->host_allocs, &plan, cells16);
    if (rc != KAS_E_OK) return rc;
    victim->plan = plan;
  }
  victim->desc.resize(desc_bytes);
  {
    size_t off = 0;
    for (int i = 0; i < 5; ++i) { if (seg_bytes[i]) memcpy(victim->desc.data() + off, seg[i], seg_bytes[i]); off += seg_bytes[i]; }
  }
  victim->key = key; victim->sig = sig; victim->last_use = ctx->use_clock; victim->call = ctx->host_calls;
  // a host call blocks until its results are back: its solve has the GPU to itself (or shares it with the few other
  // scenario ranges of the same call), so the relaxation form takes double tiles whatever the batch size — the order
  // kernel of a 1000-variant what-if call 2.0 -> 1.7 ms
  // ... and its first fit runs as a wavefront of the order kernel's workgroup where that applies (kas_p4_with_order: round 6, a
  // batch of 1000 alone 2.15 ms against 2.42 ms with first fit on the fill workgroup's four wavefronts, round 5's choice — which
  // the same two bits still give every plan the new kernel does not serve: kas_split_p4 / kas_p4_with_order)
  victim->plan->flags |= KAS_FLAG_RELAX_TILES_128 | KAS_FLAG_P4_WITH_ORDER;
  *out_plan = victim->plan;
  return KAS_E_OK;
}

// grow-only device buffer of the host path
static int kas_host_buf(kas_ctx* ctx, KasBuf* b, size_t bytes) {
  if (bytes <= b->cap) return KAS_E_OK;
  // (every host call drains its streams before it returns: nothing is in flight on the old buffer)
  return kas_buf_reserve(b, bytes + bytes / 4 + 256, &ctx->host_allocs, "host-path buffer");
}

// grow-only pinned host buffer of the host path
static int kas_host_pinned(kas_ctx* ctx, KasBuf* b, size_t bytes) {
  if (bytes <= b->cap) return KAS_E_OK;
  if (b->p) (void)hipHostFree(b->p);
  b->p = nullptr; b->cap = 0;
  const size_t want = bytes + bytes / 4 + 256;
  if (hipHostMalloc(&b->p, want, hipHostMallocDefault) != hipSuccess) { b->p = nullptr; return set_error(KAS_E_NOMEM, "hipHostMalloc failed (host-path record staging)"); }
  b->cap = want;
  ctx->host_allocs += 1;
  return KAS_E_OK;
}

// one scenario range of a host call: its slice of the batch, its plan, what it moves
struct KasChain {
  int64_t lo = 0, hi = 0;                       // scenarios
  int64_t tlo = 0, thi = 0;                     // topics (absolute indices)
  std::vector<kas_scenario_desc> scen;          // rebased descriptors of the slice
  kas_batch_desc bd;
  kas_plan* plan = nullptr;
  int64_t cur_lo = 0, cur_hi = 0, out_lo = 0, out_hi = 0;
};

// The two copy streams of a host call that is cut into scenario ranges, made on first use.  They get hardware queues
// of their own: the runtime maps ordinary streams onto a pool of GPU_MAX_HW_QUEUES (default 4) queues, and an upload
// that shares a queue with the solve of another range waits behind its kernels (a copy trace of the plain path:
// uploads in bursts of three).  A stream created with a CU mask is not pooled — the mask is a property of the queue —
// and the mask here is every CU.  Only these two: a process that holds dozens of queues is time-sliced by the
// hardware scheduler (every stream of the host path on a queue of its own, two contexts alive: configs[4]'s 35 ms
// chain kernel ran 13 % slower, configs[3]'s share fell from 590k to 341k scenarios/s).
// the host path's solve streams: the first n of hstream[] exist after this
static int kas_host_solve_streams(kas_ctx* c, int n) {
  for (int i = 0; i < n && i < KAS_HOST_STREAMS; ++i)
    if (!c->hstream[i]) KAS_HIP_TRY(hipStreamCreateWithFlags(&c->hstream[i], hipStreamNonBlocking));
  return KAS_E_OK;
}

static int kas_host_copy_streams(kas_ctx* c) {
  if (c->hup && c->hdown) return KAS_E_OK;
  hipDeviceProp_t prop;
  KAS_HIP_TRY(hipGetDeviceProperties(&prop, c->device));
  const int words = (prop.multiProcessorCount + 31) / 32;
  std::vector<uint32_t> all((size_t)words, 0xffffffffu);
  if (prop.multiProcessorCount % 32) all[(size_t)words - 1] = (1u << (prop.multiProcessorCount % 32)) - 1u;
  for (hipStream_t* st : {&c->hup, &c->hdown}) {
    if (*st) continue;
    if (hipExtStreamCreateWithCUMask(st, (uint32_t)words, all.data()) != hipSuccess) {
      (void)hipGetLastError();
      *st = nullptr;
      KAS_HIP_TRY(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    }
  }
  return KAS_E_OK;
}

// ---- 16-bit cells at the host boundary (kas_solve_host16, include/kas_abi.h) --------------------------------------
// The solver kernels read and write int32 cells; a 16-bit call moves half the bytes over PCIe and converts on the device,
// where a cell costs 6 bytes of HBM traffic against the link's 2: KAS_CELL16_NONE <-> -1, every other value as it is.
// A thread takes 8 cells, the 16-byte side of its accesses aligned (the pools start at whatever cell the descriptors
// say; the cells before the first aligned group and behind the last go one by one).
__global__ __launch_bounds__(256) void kas_cells_widen_kernel(const uint16_t* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
  const int64_t head0 = (int64_t)((16u - (uint32_t)((uintptr_t)in & 15u)) & 15u) >> 1;
  const int64_t head = head0 < n ? head0 : n;
  const int64_t groups = (n - head) >> 3;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  auto one = [&](int64_t i) { const uint32_t v = in[i]; out[i] = v == 0xffffu ? -1 : (int32_t)v; };
  for (int64_t g = tid; g < groups; g += nth) {
    const int64_t at = head + (g << 3);
    const uint4 v = *reinterpret_cast<const uint4*>(in + at);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t lo = w[k] & 0xffffu, hi = w[k] >> 16;
      out[at + 2 * k] = lo == 0xffffu ? -1 : (int32_t)lo;
      out[at + 2 * k + 1] = hi == 0xffffu ? -1 : (int32_t)hi;
    }
  }
  for (int64_t i = tid; i < head; i += nth) one(i);
  for (int64_t i = head + (groups << 3) + tid; i < n; i += nth) one(i);
}

__global__ __launch_bounds__(256) void kas_cells_narrow_kernel(const int32_t* __restrict__ in, uint16_t* __restrict__ out, int64_t n) {
  const int64_t head0 = (int64_t)((16u - (uint32_t)((uintptr_t)out & 15u)) & 15u) >> 1;
  const int64_t head = head0 < n ? head0 : n;
  const int64_t groups = (n - head) >> 3;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  // (a node index is below 32,768 and the pad value is -1: the low half of the cell is the 16-bit cell)
  for (int64_t g = tid; g < groups; g += nth) {
    const int64_t at = head + (g << 3);
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      w[k] = ((uint32_t)in[at + 2 * k] & 0xffffu) | ((uint32_t)in[at + 2 * k + 1] << 16);
    *reinterpret_cast<uint4*>(out + at) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  for (int64_t i = tid; i < head; i += nth) out[i] = (uint16_t)in[i];
  for (int64_t i = head + (groups << 3) + tid; i < n; i += nth) out[i] = (uint16_t)in[i];
}

static unsigned kas_cells_grid(int64_t n) {
  const int64_t g = (n / 8 + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

// the 16-bit side of a host call (nullptr members: the int32 call)
struct KasCells16 {
  const uint16_t* cur = nullptr;
  uint16_t* out = nullptr;
};

#define KAS_HOST_SPLIT_MIN_BYTES (48ll << 20)   // tables smaller than this are moved and solved as one range
// Scenario ranges of one call.  A range's solve is a chain of ~2.5-3 ms however few scenarios it holds (one workgroup per
// scenario walks its 100,000 rows), so a call lasts upload + that chain + the last range's download, and more ranges only
// shorten the last download — while the runtime maps the ranges' solve streams onto GPU_MAX_HW_QUEUES (default 4) hardware
// queues shared with everything else, where ranges on one queue run one after the other (a kernel + copy trace of eight
// ranges: the last one solved alone, 3.4 ms after the others).  240 scenarios x 100,000 x 3 cells, 16-bit cells, pinned:
// 2 ranges 6.4 ms, 3: 6.0, 4: 7.1 (5.9 with eight hardware queues), 8: 7.8; int32 cells 9.3-10.0 ms whatever the count
// (scripts/e2e_host_path.py; KAS_HOST_RANGES overrides for such measurements).
#define KAS_HOST_SPLIT_MAX 3

static int kas_solve_host_locked(kas_ctx* ctx, const kas_batch_desc* batch, const kas_tables* h, const int32_t* select,
                                 int32_t n_select, const KasCells16* c16 = nullptr) {
  KAS_HIP_TRY(hipSetDevice(ctx->device));
  ctx->host_calls += 1;
  const bool all_rows = n_select < 0;
  if (!all_rows && n_select > 0 && !select) return set_error(KAS_E_INVALID_ARG, "select == NULL");
  // 16-bit cells are node indices: the node table the kernels see gives node i the id i (h->cur / h->out are unused)
  kas_batch_desc ident_batch;
  if (c16) {
    const int irc = kas_ident_batch(batch, &ctx->ident_ids, &ident_batch, &ctx->ident_stamp);
    if (irc != KAS_E_OK) return irc;
    batch = &ident_batch;
  }
  const size_t cell = c16 ? sizeof(uint16_t) : sizeof(int32_t);     // bytes of a cur / out cell as it travels
  const bool have_cur = c16 ? c16->cur != nullptr : h->cur != nullptr;
  const bool have_out = c16 ? c16->out != nullptr : h->out != nullptr;
  // the whole batch: validation and the extents of every pool
  KasShape full;
  {
    std::string err;
    const int rc = kas_shape_batch(batch, &full, &err, 0, 0);
    if (rc != KAS_E_OK) return set_error(rc, err);
  }
  const int64_t S = batch->n_scenarios, T = batch->n_topics;
  // A 16-bit call is solved on its 16-bit cells where the kernels with that I/O take the batch (what kas_plan_create16
  // accepts: lists up to 3 wide, relaxation or round form — kas_plan_build's test, made HERE on the shape: a batch the
  // 16-bit kernels refuse never reaches the plan cache with the 16-bit bit set); any other batch is widened
  // before and narrowed behind an int32 solve.
  bool native16 = c16 != nullptr && full.Wc <= 3 &&
                  ((full.relax_ok && ctx->lds_lane_order_ok && kas_order_relax_for(full.Wc, 0, 0, 0, 1) != nullptr) || full.round_fits);
  const bool need32 = c16 == nullptr || !native16;             // int32 cell pools on the device
  if (h->cur_len < full.cur_need || (all_rows && h->out_len < full.out_need) || h->aux_len < full.aux_need ||
      h->ctx_len < full.ctx_need)
    return set_error(KAS_E_INVALID_ARG, "a descriptor offset reaches beyond the pool length given in kas_tables");
  if ((full.cur_need > full.cur_lo && !have_cur) || (all_rows && full.out_need > full.out_lo && !have_out) ||
      (full.aux_need > full.aux_lo && !h->aux) || (full.ctx_need > full.ctx_lo && !h->ctx) || (T && !h->topic_results) ||
      (S && !h->scenario_results))
    return set_error(KAS_E_INVALID_ARG, "a table the descriptors refer to is NULL");
  // rows of the selected scenarios, packed: where each one goes
  std::vector<int64_t> sel_off;
  if (!all_rows) {
    int64_t at = 0;
    sel_off.resize((size_t)n_select + 1);
    for (int32_t k = 0; k < n_select; ++k) {
      if (select[k] < 0 || select[k] >= S) return set_error(KAS_E_INVALID_ARG, "select: scenario index out of range");
      sel_off[(size_t)k] = at;
      const kas_scenario_desc& sd = batch->scenarios[select[k]];
      for (int32_t t = 0; t < sd.topic_count; ++t) {
        const kas_topic_desc& td = batch->topics[sd.topic_begin + t];
        at += (int64_t)td.n_partitions * td.out_width;
      }
    }
    sel_off[(size_t)n_select] = at;
    if (h->out_len < at || (at > 0 && !have_out))
      return set_error(KAS_E_INVALID_ARG, "select: out / out_len too small for the selected scenarios' rows");
  }
  // device pools: only [lo, need) of each is ever touched, the pointers are rebased so that the
  // descriptors' absolute offsets apply (+ 8 ints: the fill kernel's full-row loads re-read the last
  // row for lanes past the end)
  int rc;
  // (a 16-bit call solved on its own cells never touches the int32 pools: not reserved for it)
  if (need32 && (rc = kas_host_buf(ctx, &ctx->h_cur, sizeof(int32_t) * (size_t)(full.cur_need - full.cur_lo + 8))) != KAS_E_OK) return rc;
  if (need32 && (rc = kas_host_buf(ctx, &ctx->h_out, sizeof(int32_t) * (size_t)(full.out_need - full.out_lo + 8))) != KAS_E_OK) return rc;
  if ((rc = kas_host_buf(ctx, &ctx->h_aux, sizeof(int32_t) * (size_t)(full.aux_need - full.aux_lo + 8))) != KAS_E_OK) return rc;
  if ((rc = kas_host_buf(ctx, &ctx->h_ctx, sizeof(int32_t) * (size_t)(full.ctx_need - full.ctx_lo + 8))) != KAS_E_OK) return rc;
  if (c16 && ((rc = kas_host_buf(ctx, &ctx->h_cur16, sizeof(uint16_t) * (size_t)(full.cur_need - full.cur_lo + 8))) != KAS_E_OK ||
              (rc = kas_host_buf(ctx, &ctx->h_out16, sizeof(uint16_t) * (size_t)(full.out_need - full.out_lo + 8))) != KAS_E_OK))
    return rc;
  if ((rc = kas_host_buf(ctx, &ctx->h_tr, sizeof(kas_topic_result) * (size_t)(T + 1))) != KAS_E_OK) return rc;
  if ((rc = kas_host_buf(ctx, &ctx->h_sr, sizeof(kas_scenario_result) * (size_t)(S + 1))) != KAS_E_OK) return rc;
  // The result records come back through pinned staging of the context's own: the caller's record arrays are ordinary
  // (pageable) memory even when its bulk tables are pinned, and a copy to pageable memory blocks the issuing thread
  // until the stream gets there — every range's records would hold the next range's upload back until its own solve
  // has finished (a memory-copy trace of the plain path showed exactly that: uploads 2-3 ms apart).
  if ((rc = kas_host_pinned(ctx, &ctx->h_tr_pin, sizeof(kas_topic_result) * (size_t)(T + 1))) != KAS_E_OK) return rc;
  if ((rc = kas_host_pinned(ctx, &ctx->h_sr_pin, sizeof(kas_scenario_result) * (size_t)(S + 1))) != KAS_E_OK) return rc;
  kas_topic_result* p_tr = (kas_topic_result*)ctx->h_tr_pin.p;
  kas_scenario_result* p_sr = (kas_scenario_result*)ctx->h_sr_pin.p;
  int32_t* d_cur = need32 ? (int32_t*)ctx->h_cur.p - full.cur_lo : nullptr;
  int32_t* d_out = need32 ? (int32_t*)ctx->h_out.p - full.out_lo : nullptr;
  uint16_t* d_cur16 = c16 ? (uint16_t*)ctx->h_cur16.p - full.cur_lo : nullptr;
  uint16_t* d_out16 = c16 ? (uint16_t*)ctx->h_out16.p - full.out_lo : nullptr;
  int32_t* d_aux = (int32_t*)ctx->h_aux.p - full.aux_lo;
  int32_t* d_ctx = (int32_t*)ctx->h_ctx.p - full.ctx_lo;
  kas_topic_result* d_tr = (kas_topic_result*)ctx->h_tr.p;
  kas_scenario_result* d_sr = (kas_scenario_result*)ctx->h_sr.p;

  // ---- scenario ranges (chains).  Large tables laid out scenario by scenario are cut so that the upload
  // of one range, the solve of the previous one and the download of the one before overlap.
  const int64_t bytes_in = (int64_t)cell * (full.cur_need - full.cur_lo);
  const int64_t bytes_out = all_rows ? (int64_t)cell * (full.out_need - full.out_lo) : 0;
  int K = 1;
  if (bytes_in + bytes_out >= KAS_HOST_SPLIT_MIN_BYTES && S >= 2) {
    K = (int)((bytes_in + bytes_out) / (KAS_HOST_SPLIT_MIN_BYTES / 2));
    if (K > KAS_HOST_SPLIT_MAX) K = KAS_HOST_SPLIT_MAX;
    if (const char* e = getenv("KAS_HOST_RANGES")) { const int k = atoi(e); if (k >= 1 && k <= KAS_HOST_STREAMS) K = k; }
    if (K > S) K = (int)S;
  }
  std::vector<KasChain> chains;
  for (int attempt = 0; attempt < 2; ++attempt) {
    chains.assign((size_t)K, KasChain());
    bool ok = true;
    for (int i = 0; i < K && ok; ++i) {
      KasChain& c = chains[(size_t)i];
      kas_shard_range(S, i, K, &c.lo, &c.hi);
      c.scen.resize((size_t)(c.hi - c.lo) + 1);
      if ((rc = kas_batch_slice(batch, c.lo, c.hi, c.scen.data(), &c.bd, nullptr, nullptr)) != KAS_E_OK) return rc;
      c.tlo = c.bd.topics ? (int64_t)(c.bd.topics - batch->topics) : 0;
      c.thi = c.tlo + c.bd.n_topics;
      if (K > 1) {
        // extents of the range; ranges must own disjoint, ascending stretches of topics, cur and out
        KasShape sh;
        std::string err;
        if ((rc = kas_shape_batch(&c.bd, &sh, &err, 0, 0)) != KAS_E_OK) return set_error(rc, err);
        c.cur_lo = sh.cur_lo; c.cur_hi = sh.cur_need; c.out_lo = sh.out_lo; c.out_hi = sh.out_need;
        if (i > 0) {
          const KasChain& q = chains[(size_t)i - 1];
          ok = c.tlo >= q.thi && c.cur_lo >= q.cur_hi && c.out_lo >= q.out_hi;
        }
      } else {
        c.cur_lo = full.cur_lo; c.cur_hi = full.cur_need; c.out_lo = full.out_lo; c.out_hi = full.out_need;
      }
    }
    if (ok) break;
    K = 1;                                                     // shared or interleaved tables: one range
  }
  for (KasChain& c : chains) {
    rc = kas_host_plan(ctx, &c.bd, &c.plan, native16 ? 1 : 0);
    if (rc == KAS_E_UNSUPPORTED && native16)                   // (cannot happen: a range's shape is a part of the batch's)
      return set_error(KAS_E_UNSUPPORTED, "internal: a scenario range of a batch the 16-bit kernels take was refused by them: " + g_last_error);
    if (rc != KAS_E_OK) return rc;
  }

  // ---- enqueue.  From here on every exit drains the streams first.
  if ((rc = kas_host_solve_streams(ctx, K > 1 ? K : 1)) != KAS_E_OK) return rc;
  hipStream_t s0 = ctx->hstream[0];
  hipError_t he = hipSuccess;
  int fail_rc = KAS_E_OK;
  auto hip_ok = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && he == hipSuccess) { he = e; fail_rc = set_error(KAS_E_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
    return he == hipSuccess;
  };
  // every stream is drained whatever happened; an error that only surfaces at a synchronisation (a kernel fault, a
  // failed copy) is the call's error: the caller must never read out / ctx / records of a solve that did not finish
  if (K > 1 && (rc = kas_host_copy_streams(ctx)) != KAS_E_OK) return rc;
  auto drain = [&]() {
    for (hipStream_t st : ctx->hstream) if (st) hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (ctx->hup) hip_ok(hipStreamSynchronize(ctx->hup), "hipStreamSynchronize");
    if (ctx->hdown) hip_ok(hipStreamSynchronize(ctx->hdown), "hipStreamSynchronize");
  };
  static_assert(KAS_HOST_SPLIT_MAX <= KAS_HOST_STREAMS, "one event pair per scenario range");
  // one range: everything on s0.  Several: uploads on hup, solves on hstream[i], downloads on hdown.
  hipStream_t s_up = K > 1 ? ctx->hup : s0, s_down = K > 1 ? ctx->hdown : s0;
  // pools every range reads (aux, Context) go up first (the ranges' own uploads queue behind them on the same stream)
  if (full.aux_need > full.aux_lo)
    hip_ok(hipMemcpyAsync(d_aux + full.aux_lo, h->aux + full.aux_lo, 4 * (size_t)(full.aux_need - full.aux_lo), hipMemcpyHostToDevice, s_up), "upload aux");
  if (full.ctx_need > full.ctx_lo)
    hip_ok(hipMemcpyAsync(d_ctx + full.ctx_lo, h->ctx + full.ctx_lo, 4 * (size_t)(full.ctx_need - full.ctx_lo), hipMemcpyHostToDevice, s_up), "upload ctx");
  auto download = [&](int i) {
    const KasChain& c = chains[(size_t)i];
    if (K > 1) hip_ok(hipStreamWaitEvent(s_down, ctx->hev_done[i], 0), "wait");
    if (all_rows && c.out_hi > c.out_lo) {
      if (c16) hip_ok(hipMemcpyAsync(c16->out + c.out_lo, d_out16 + c.out_lo, 2 * (size_t)(c.out_hi - c.out_lo), hipMemcpyDeviceToHost, s_down), "download out");
      else hip_ok(hipMemcpyAsync(h->out + c.out_lo, d_out + c.out_lo, 4 * (size_t)(c.out_hi - c.out_lo), hipMemcpyDeviceToHost, s_down), "download out");
    }
    if (c.thi > c.tlo)
      hip_ok(hipMemcpyAsync(p_tr + c.tlo, d_tr + c.tlo, sizeof(kas_topic_result) * (size_t)(c.thi - c.tlo), hipMemcpyDeviceToHost, s_down), "download topic results");
    if (c.hi > c.lo)
      hip_ok(hipMemcpyAsync(p_sr + c.lo, d_sr + c.lo, sizeof(kas_scenario_result) * (size_t)(c.hi - c.lo), hipMemcpyDeviceToHost, s_down), "download scenario results");
  };
  // software-pipelined issue order (upload i, solve i, download i - 2): with pageable host memory the copies block
  // the issuing thread, and this order still lets the device overlap them with the solves.  (A range's solve takes
  // about as long as three uploads: behind upload i the solve of range i - 1 is still running, that of i - 2 is done —
  // a lag of one made the thread wait ~1 ms per range: 15.6 ms per call of eight ranges.)
  const int lag = K >= 3 ? 2 : 1;
  for (int i = 0; i < K && he == hipSuccess && fail_rc == KAS_E_OK; ++i) {
    KasChain& c = chains[(size_t)i];
    hipStream_t st = K > 1 ? ctx->hstream[i % KAS_HOST_STREAMS] : s0;
    if (c.cur_hi > c.cur_lo) {
      if (c16) hip_ok(hipMemcpyAsync(d_cur16 + c.cur_lo, c16->cur + c.cur_lo, 2 * (size_t)(c.cur_hi - c.cur_lo), hipMemcpyHostToDevice, s_up), "upload cur");
      else hip_ok(hipMemcpyAsync(d_cur + c.cur_lo, h->cur + c.cur_lo, 4 * (size_t)(c.cur_hi - c.cur_lo), hipMemcpyHostToDevice, s_up), "upload cur");
    }
    if (K > 1) {
      hip_ok(hipEventRecord(ctx->hev_up[i], s_up), "record");
      hip_ok(hipStreamWaitEvent(st, ctx->hev_up[i], 0), "wait");
    }
    if (he != hipSuccess) break;
    if (c16 && !native16 && c.cur_hi > c.cur_lo) {             // (on the range's solve stream: the copy streams only copy)
      const int64_t n = c.cur_hi - c.cur_lo;
      hipLaunchKernelGGL(kas_cells_widen_kernel, dim3(kas_cells_grid(n)), dim3(256), 0, st, d_cur16 + c.cur_lo, d_cur + c.cur_lo, n);
      if (!hip_ok(hipGetLastError(), "kas_cells_widen_kernel")) break;
    }
    kas_tables d;
    memset(&d, 0, sizeof(d));
    d.cur = d_cur; d.out = d_out; d.aux = d_aux; d.ctx = d_ctx;
    d.topic_results = d_tr + c.tlo; d.scenario_results = d_sr + c.lo;
    if (native16) { d.cur = reinterpret_cast<const int32_t*>(d_cur16); d.out = reinterpret_cast<int32_t*>(d_out16); }
    const int src = kas_solve_device_impl(c.plan, &d, st);
    if (src != KAS_E_OK) { fail_rc = src; break; }
    if (c16 && !native16 && all_rows && c.out_hi > c.out_lo) {
      const int64_t n = c.out_hi - c.out_lo;
      // (into the staging buffer, not into the caller's pinned pool: the kernel's stores over the link ran at 20 GB/s —
      // a call of 240 scenarios 10.5 ms against 8.4 ms with the copy engine, experiments/README.md)
      hipLaunchKernelGGL(kas_cells_narrow_kernel, dim3(kas_cells_grid(n)), dim3(256), 0, st, d_out + c.out_lo, d_out16 + c.out_lo, n);
      if (!hip_ok(hipGetLastError(), "kas_cells_narrow_kernel")) break;
    }
    if (K > 1) hip_ok(hipEventRecord(ctx->hev_done[i], st), "record");
    if (i >= lag) download(i - lag);
  }
  for (int i = K - lag < 0 ? 0 : K - lag; i < K && he == hipSuccess && fail_rc == KAS_E_OK; ++i) download(i);
  drain();
  if (he != hipSuccess || fail_rc != KAS_E_OK) return fail_rc != KAS_E_OK ? fail_rc : KAS_E_HIP;
  for (const KasChain& c : chains) {                           // (only what the ranges own: as the direct copies did)
    if (c.thi > c.tlo) memcpy(h->topic_results + c.tlo, p_tr + c.tlo, sizeof(kas_topic_result) * (size_t)(c.thi - c.tlo));
    if (c.hi > c.lo) memcpy(h->scenario_results + c.lo, p_sr + c.lo, sizeof(kas_scenario_result) * (size_t)(c.hi - c.lo));
  }
  // Context counters back; the selected scenarios' rows, packed
  if (full.ctx_need > full.ctx_lo)
    hip_ok(hipMemcpyAsync(h->ctx + full.ctx_lo, d_ctx + full.ctx_lo, 4 * (size_t)(full.ctx_need - full.ctx_lo), hipMemcpyDeviceToHost, s0), "download ctx");
  if (!all_rows) {
    for (int32_t k = 0; k < n_select && he == hipSuccess; ++k) {
      const kas_scenario_desc& sd = batch->scenarios[select[k]];
      int64_t at = sel_off[(size_t)k];
      for (int32_t t = 0; t < sd.topic_count && he == hipSuccess; ++t) {
        const kas_topic_desc& td = batch->topics[sd.topic_begin + t];
        const int64_t cells = (int64_t)td.n_partitions * td.out_width;
        if (cells > 0 && c16) {
          if (!native16) {
            hipLaunchKernelGGL(kas_cells_narrow_kernel, dim3(kas_cells_grid(cells)), dim3(256), 0, s0, d_out + td.out_off, d_out16 + td.out_off, cells);
            hip_ok(hipGetLastError(), "kas_cells_narrow_kernel");
          }
          hip_ok(hipMemcpyAsync(c16->out + at, d_out16 + td.out_off, 2 * (size_t)cells, hipMemcpyDeviceToHost, s0), "download selected rows");
        } else if (cells > 0)
          hip_ok(hipMemcpyAsync(h->out + at, d_out + td.out_off, 4 * (size_t)cells, hipMemcpyDeviceToHost, s0), "download selected rows");
        at += cells;
      }
    }
  }
  hip_ok(hipStreamSynchronize(s0), "hipStreamSynchronize");
  return he == hipSuccess ? KAS_E_OK : fail_rc;
}

int kas_solve_host_select(kas_ctx* ctx, const kas_batch_desc* batch, const kas_tables* h, const int32_t* select, int32_t n_select) {
  if (!ctx || !batch || !h) return set_error(KAS_E_INVALID_ARG, "NULL argument");
  std::lock_guard<std::mutex> lock(ctx->host_mu);
  return kas_solve_host_locked(ctx, batch, h, select, n_select);
}

int kas_solve_host(kas_ctx* ctx, const kas_batch_desc* batch, const kas_tables* h) {
  return kas_solve_host_select(ctx, batch, h, nullptr, -1);
}

int kas_solve_host16(kas_ctx* ctx, const kas_batch_desc* batch, const kas_tables16* h16, const int32_t* select, int32_t n_select) {
  if (!ctx || !batch || !h16) return set_error(KAS_E_INVALID_ARG, "NULL argument");
  kas_tables h;                                                // everything but the cells is the int32 call's
  memset(&h, 0, sizeof(h));
  h.aux = h16->aux; h.ctx = h16->ctx; h.topic_results = h16->topic_results; h.scenario_results = h16->scenario_results;
  h.cur_len = h16->cur_len; h.out_len = h16->out_len; h.aux_len = h16->aux_len; h.ctx_len = h16->ctx_len;
  KasCells16 c16;
  c16.cur = h16->cur; c16.out = h16->out;
  std::lock_guard<std::mutex> lock(ctx->host_mu);
  return kas_solve_host_locked(ctx, batch, &h, select, n_select < 0 ? -1 : n_select, &c16);
}

int kas_solve_host_sharded(kas_ctx* const* ctxs, int32_t n_ctx, const kas_batch_desc* batch, const kas_tables* h) {
  if (!ctxs || n_ctx < 1 || !batch || !h) return set_error(KAS_E_INVALID_ARG, "NULL argument / no context");
  for (int32_t r = 0; r < n_ctx; ++r) if (!ctxs[r]) return set_error(KAS_E_INVALID_ARG, "ctxs[r] == NULL");
  if (n_ctx == 1) return kas_solve_host(ctxs[0], batch, h);
  // Every shard downloads the whole extent of `out` (and round-trips the whole extent of `ctx`) its scenarios refer
  // to: with pools that are not laid out in scenario order one shard's download would overwrite rows or Context
  // counters another has already delivered.  Such a batch is solved on one context (same results, no overlap).
  {
    int64_t out_end = -1, ctx_end = -1;
    bool disjoint = true;
    for (int32_t r = 0; r < n_ctx && disjoint; ++r) {
      int64_t lo = 0, hi = 0;
      kas_shard_range(batch->n_scenarios, r, n_ctx, &lo, &hi);
      if (hi <= lo) continue;
      std::vector<kas_scenario_desc> scratch((size_t)(hi - lo));
      kas_batch_desc bd;
      kas_tables ht;
      KasShape sh;
      std::string err;
      int rc = kas_batch_slice(batch, lo, hi, scratch.data(), &bd, h, &ht);
      if (rc == KAS_E_OK) rc = kas_shape_batch(&bd, &sh, &err, 0, 0);
      if (rc != KAS_E_OK) return rc == KAS_E_OK ? rc : set_error(rc, "shard " + std::to_string(r) + ": " + (err.empty() ? g_last_error : err));
      if (sh.out_need > sh.out_lo) { disjoint = disjoint && sh.out_lo >= out_end; out_end = sh.out_need; }
      if (sh.ctx_need > sh.ctx_lo) { disjoint = disjoint && sh.ctx_lo >= ctx_end; ctx_end = sh.ctx_need; }
    }
    if (!disjoint) return kas_solve_host(ctxs[0], batch, h);
  }
  std::vector<int> rcs((size_t)n_ctx, KAS_E_OK);
  std::vector<std::string> errs((size_t)n_ctx);
  std::vector<std::thread> threads;
  for (int32_t r = 0; r < n_ctx; ++r) {
    threads.emplace_back([&, r]() {
      int64_t lo = 0, hi = 0;
      kas_shard_range(batch->n_scenarios, r, n_ctx, &lo, &hi);
      if (hi <= lo) return;
      std::vector<kas_scenario_desc> scratch((size_t)(hi - lo));
      kas_batch_desc bd;
      kas_tables ht;
      int rc = kas_batch_slice(batch, lo, hi, scratch.data(), &bd, h, &ht);
      if (rc == KAS_E_OK) rc = kas_solve_host(ctxs[r], &bd, &ht);
      rcs[(size_t)r] = rc;
      if (rc != KAS_E_OK) errs[(size_t)r] = g_last_error;      // (thread-local: carried to the caller below)
    });
  }
  for (std::thread& t : threads) t.join();
  for (int32_t r = 0; r < n_ctx; ++r)
    if (rcs[(size_t)r] != KAS_E_OK) return set_error(rcs[(size_t)r], "shard " + std::to_string(r) + ": " + errs[(size_t)r]);
  return KAS_E_OK;
}

int kas_ctx_host_stats(kas_ctx* ctx, int64_t* calls, int64_t* plan_hits, int64_t* device_allocs) {
  if (!ctx) return set_error(KAS_E_INVALID_ARG, "ctx == NULL");
  std::lock_guard<std::mutex> lock(ctx->host_mu);
  if (calls) *calls = (int64_t)ctx->host_calls;
  if (plan_hits) *plan_hits = (int64_t)ctx->host_plan_hits;
  if (device_allocs) *device_allocs = (int64_t)ctx->host_allocs;
  return KAS_E_OK;
}

}  // extern "C"
#pragma GCC visibility pop
