// kas_solver_body.h — the per-scenario solver: device code of the fill kernel (one workgroup of
// NW 64-lane wavefronts per scenario, the scenario's node state in LDS) and of the order kernels.
//
// Computes exactly what KafkaAssignmentStrategy.getRackAwareAssignment computes
// (KafkaAssignmentStrategy.java:40-63, "KAS"), for every topic of a scenario in order, with the
// pre-checks of KafkaTopicAssigner.generateAssignment (KafkaTopicAssigner.java:65-69, "KTA").
// The reference is sequential; each phase below is an order-preserving parallel formulation
// whose result is identical to the sequential one (DESIGN.md "Exact parallel formulations"):
//
//  P0  cap = (int)ceil((double)(int)(P*rf)/N)                                     KAS:65-71
//  P2  sticky fill (KAS:101-131), rack-diverse form.  When every row's valid replicas sit on
//      pairwise different racks the rack test of canAccept can never fail during the fill, so a
//      node keeps its first `cap` candidates in (replica index, row) order independently of
//      every other node.  The row range is cut into NW chunks, one per wavefront:
//        A   wave w scans chunk w: per-chunk histogram hist[n][w][r] = candidates of sweep r on
//            node n (uint16 cells, LDS atomics) and the proof of rack diversity
//        Q   per node: the one sweep r* in which it saturates, its quota q there, and — a prefix
//            over the node's own per-chunk counts — the quota left when chunk w starts
//            (lists wider than 3, or tables that do not fit: one histogram for the whole topic,
//            then a separate pass A2 in which wave w counts chunk w's sweep-r* candidates)
//        B   wave w walks chunk w in row order: replica (p, r) on n is kept iff r < r*(n), or
//            r == r*(n) and its rank among n's sweep-r* candidates is below the quota (ranked
//            by ballot only in the one tile where the quota runs out).  P3 (KAS:133-160) runs
//            in the same scan: holders -> mid row (uint16 node indices at the end of the topic's
//            out region), movement counts, and the orphan rows appended to the chunk's list in
//            HBM scratch (ascending).
//      General form (rows that are not rack-diverse, or no LDS for the histogram): one sweep per
//      replica index over 64-row tiles by wave 0, accept masks as ballot words in HBM scratch.
//  P4  first fit (KAS:162-186): orphans in ascending row order, 64 per window, evaluated
//      position-major over the compacted list of non-full nodes in processing order (a full
//      node never becomes non-full; the reference spends >99% of its probes on them); the windows
//      go round-robin to all wavefronts, one step apart.
//  P5  preference order (KAS:202-239), a kernel of its own (the order kernel): its only state
//      is count[node][replica index], so it runs with a fraction of the fill kernel's LDS and
//      many more scenarios per CU.  Row p reads count[n][0..L) of its own nodes, picks, then
//      increments one counter per node, so it only has to wait for the EARLIER rows that hold
//      one of its nodes.  The chain of such waits is long (every orphan placed by first fit
//      lands on the same few nodes) and only ~10-20 rows are ever ready at once: P5 is
//      latency-bound per scenario, and throughput comes from running many scenarios side by side.
//      Ticket form (order_tickets, lists <= 3 wide; kas_order_wide.h, lists 4 and 5 wide): a
//      staging wavefront streams the mid rows and hands every (row, node) its ticket = how many
//      earlier rows of the scenario hold that node.  Every committed row adds exactly 1 to the
//      commit count of each of its nodes, so "commits on n == ticket" says that all earlier rows
//      on n have committed: the solving wavefront's lanes claim rows in order, wait on their
//      tickets, pick and commit with one LDS atomic add per node — no tile-wide rounds, rows that
//      queue on one node are decided together, and a wavefront can serve several scenarios at
//      once (one lane group each); a retiring wavefront writes the final rows.
//      Round form (Context handed in, lists wider than 5, ticket overflow, the KAS:190 index
//      error): one wavefront per scenario, 64 ascending rows per tile, a lane commits once no
//      lower lane sharing a node is pending.
//
// Everything cross-lane goes through kas_wave.h; control flow around wave collectives is
// wave-uniform and around kasw::sync() workgroup-uniform.  List positions live in registers
// through fully unrolled loops (W is a template parameter).
#pragma once
#include <stdint.h>

#include <type_traits>

#include "kas_abi.h"
#include "kas_plan_math.h"
#include "kas_wave.h"

// Event counters of the fill kernel (P4 windows / node steps, ranked tiles; kas_plan_stats() [4],
// [5], [7]) and the solver's "rows in hand" tally ([15]).  They are per-lane 64-bit values that live
// across a whole kernel: in the product build they are compiled out (zero) — the row scans are
// short of registers and ran 20-35 % slower with them (round 2, profiles/) — and a
// -DKAS_FILL_COUNTERS build brings them back.
#ifdef KAS_FILL_COUNTERS
#define KAS_COUNT(x) do { (x) += 1; } while (0)
#define KAS_COUNTERS_ON 1
#else
#define KAS_COUNT(x) do { } while (0)
#define KAS_COUNTERS_ON 0
#endif

// Hang containment.  The order kernels' wavefronts and the P4 windows of the fill kernel wait for each other by
// polling LDS words; a protocol error there would spin forever and take the GPU with it.  Every such loop is
// bounded: a wavefront that polls KAS_SPIN_BOUND times without making progress raises the workgroup's watchdog
// word, every polling loop of the workgroup leaves when it sees the word, and the scenario is reported as
// KAS_FAIL_WATCHDOG instead of hanging.  Product build (round 3): 2^25 polls — 1.3 s of the tightest loop (a P4
// window waiting for its predecessors), ~15 s of a solver's; no wait between resident wavefronts of one workgroup
// lasts a millisecond — and the word is read on every 4096th idle poll only, which costs nothing measurable
// (362.8k against 362.9k scenarios/s, three A/B pairs).  Test builds use small bounds (-DKAS_SPIN_BOUND=n, below
// 65536: checked on every poll); -DKAS_SPIN_BOUND=0 compiles the containment out.
// Round 4: the wide ticket form carries the bound in the product build as well (rounds 1-3: test builds only — two
// more VGPRs and 1.4 % on its chain solver's step, the critical path of configs[4]: a protocol slip there was a hung
// GPU instead of a status, and that kernel is the most intricate one; the step's three LDS round trips removed in the
// same round paid for it three times over).  The relaxation form has no wavefront waiting for another; its loop is
// bounded by construction (65 evaluations per tile) and reports KAS_FAIL_WATCHDOG if it ever is not.
#ifndef KAS_SPIN_BOUND
#define KAS_SPIN_BOUND (1 << 25)
#endif
#ifndef KAS_WIDE_SPIN_BOUND
#define KAS_WIDE_SPIN_BOUND KAS_SPIN_BOUND
#endif
#ifndef KAS_SPIN_CHECK
#define KAS_SPIN_CHECK 4096
#endif
// test hook of the debug build: the staging wavefront stops handing out rows after this many tiles
#ifndef KAS_TEST_STALL_AFTER
#define KAS_TEST_STALL_AFTER 0
#endif

namespace kas {

// one poll of a bounded loop: `progress` (wave-uniform) = this iteration did something; returns
// true when the loop must be left (watchdog raised by this or another wavefront)
template <int BOUND = KAS_SPIN_BOUND>
KAS_DEV bool watchdog_poll(uint32_t* wd, bool progress, int32_t& idle) {
  if constexpr (BOUND > 0) {
    if (progress) { idle = 0; return false; }             // (wave-uniform; nothing else on the path of a step that did something)
    idle = kasw::uniform(idle + 1);
    // a large bound: the word is looked at on every KAS_SPIN_CHECK-th poll without progress only (a wavefront
    // that keeps making progress never looks; it stalls soon enough if the workgroup is stuck)
    constexpr int32_t every = BOUND >= 65536 ? KAS_SPIN_CHECK : 1;
    if (every > 1 && (idle & (every - 1)) != every - 1) return false;
    if (idle > BOUND && kasw::lane() == 0) *(volatile uint32_t*)wd = 1u;
    kasw::repoll();
    return kasw::ballot(*(volatile uint32_t*)wd != 0u) != 0ull;
  } else {
    (void)wd; (void)progress; (void)idle;
    return false;
  }
}

// for a wavefront that obeys the watchdog without counting for it: `polls` = its idle polls so far
template <int BOUND = KAS_SPIN_BOUND>
KAS_DEV bool watchdog_raised(uint32_t* wd, int32_t polls) {
  if constexpr (BOUND > 0) {
    constexpr int32_t every = BOUND >= 65536 ? KAS_SPIN_CHECK : 1;
    if (every > 1 && (kasw::uniform(polls) & (every - 1)) != 0) return false;
    kasw::repoll();
    return kasw::ballot(*(volatile uint32_t*)wd != 0u) != 0ull;
  } else {
    (void)wd; (void)polls;
    return false;
  }
}

struct TopicOutcome {
  int32_t status;
  int32_t fail_partition;
  int32_t moved_replicas;
  int32_t moved_partitions;
};

template <int D>
struct RowWords { uint32_t v[D]; };

// LDS of the fill kernel (kas_fill_lds_layout)
struct LdsView {
  int32_t* x;           // hist[W][N], then qc[NW][N]
  int32_t* load;        // node state: words of node n at load[n * ns], qrs[n * ns], rack[n * rs] (lds_load / lds_qrs /
  int32_t* qrs;         // lds_rack).  Dense arrays (ns = rs = 1), or — fused histogram layout — words of the node's
  int16_t* rack;        // own block in x (ns = block words, rs = 2 ns): see kas_fill_lds_layout
  int32_t ns, rs;
  int16_t* live;
  int16_t* idmap;
  int32_t* ids;
  int32_t* ring_p;
  int32_t* ring_meta;
  int16_t* ring_rack;   // [W][KAS_RING_CAP]
  int32_t* ctl;
};

KAS_DEV int32_t& lds_load(const LdsView& L, int32_t n) { return L.load[kasw::mul24(n, L.ns)]; }
KAS_DEV int32_t& lds_qrs(const LdsView& L, int32_t n) { return L.qrs[kasw::mul24(n, L.ns)]; }
KAS_DEV int16_t& lds_rack(const LdsView& L, int32_t n) { return L.rack[kasw::mul24(n, L.rs)]; }

struct NodeMap {
  int32_t n;            // N
  int32_t min_id;
  uint32_t range;       // direct-table extent, 0 = binary search
};

// nodeMap.get(nodeId) (KAS:119): node index of a broker id, or -1.
// The three row scans of the rack-diverse fill are instantiated per lookup mode (DIRECT: the
// broker-id range fits the LDS table, the usual case; else binary search over the sorted ids) and
// the mode is chosen once per topic: the search loop, unrolled per list position and tile in
// flight, would otherwise sit in the hot loops' register budget.
template <bool DIRECT>
KAS_DEV int32_t node_lookup_as(const LdsView& L, const NodeMap& m, int32_t id) {
  if (DIRECT) {
    uint32_t d = (uint32_t)id - (uint32_t)m.min_id;
    return d < m.range ? (int32_t)L.idmap[d] : -1;
  }
  int32_t lo = 0, hi = m.n - 1, res = -1;
  while (lo <= hi) {
    int32_t mid = (lo + hi) >> 1;
    int32_t v = L.ids[mid];
    if (v == id) { res = mid; break; }
    if (v < id) lo = mid + 1; else hi = mid - 1;
  }
  return res;
}

KAS_DEV int32_t node_lookup(const LdsView& L, const NodeMap& m, int32_t id) {
  if (m.range != 0u) {
    uint32_t d = (uint32_t)id - (uint32_t)m.min_id;
    return d < m.range ? (int32_t)L.idmap[d] : -1;
  }
  int32_t lo = 0, hi = m.n - 1, res = -1;
  while (lo <= hi) {
    int32_t mid = (lo + hi) >> 1;
    int32_t v = L.ids[mid];
    if (v == id) { res = mid; break; }
    if (v < id) lo = mid + 1; else hi = mid - 1;
  }
  return res;
}

// getMaxReplicasPerNode (KAS:65-71): int product (wraps like Java), double divide, ceil.
KAS_DEV int32_t max_replicas_per_node(int32_t n_nodes, int32_t n_partitions, int32_t rf) {
  int32_t prod = (int32_t)((uint32_t)n_partitions * (uint32_t)rf);
  double c = __builtin_ceil((double)prod / (double)n_nodes);
  if (c >= 2147483647.0) return 2147483647;
  if (c <= -2147483648.0) return (int32_t)0x80000000;
  return (int32_t)c;
}

// Math.abs(hash) % n with Java semantics (KAS:190); negative only for Integer.MIN_VALUE.
KAS_DEV int32_t java_abs_mod(int32_t hash, int32_t n) {
  int32_t a = (hash == (int32_t)0x80000000) ? hash : (hash < 0 ? -hash : hash);
  return a % n;
}

template <int W>
KAS_DEV int32_t sel(const int32_t (&a)[W], int32_t i) {
  int32_t t[W];
#pragma unroll
  for (int j = 0; j < W; ++j) t[j] = kasw::opaque(a[j]);   // launder first, select afterwards
  int32_t v = t[0];
#pragma unroll
  for (int j = 1; j < W; ++j) v = (i == j) ? t[j] : v;
  return v;
}

template <int W>
constexpr int cnt_stride() { return W == 3 ? 4 : W; }   // == kas_cnt_stride(W)

struct alignas(16) CntRow4 { int32_t v[4]; };

// one Context counter row (count[node][0..W)) from LDS; a 3-wide row is one 16-byte read
template <int W>
KAS_DEV void load_cnt_row(int32_t (&c)[W], const int32_t* row) {
  if constexpr (W == 3 || W == 4) {
    const CntRow4 q = *reinterpret_cast<const CntRow4*>(row);
#pragma unroll
    for (int r = 0; r < W; ++r) c[r] = q.v[r];
  } else {
#pragma unroll
    for (int r = 0; r < W; ++r) c[r] = row[r];
  }
}

template <int W>
KAS_DEV void put(int32_t (&a)[W], int32_t i, int32_t v) {
#pragma unroll
  for (int j = 0; j < W; ++j) a[j] = (i == j) ? v : a[j];
}

// first tile of chunk w when nt tiles are cut into NW contiguous chunks
template <int NW>
KAS_DEV int32_t chunk_begin(int32_t nt, int32_t w) { return (int32_t)(((int64_t)nt * w) / NW); }

// ---------------------------------------------------------------------------------------------
// P4 window: up to 64 orphans (lane = orphan, ascending row order), position-major first fit.
// Returns -1, or the lane index of the first orphan that cannot be fully assigned (KAS:183).
// ---------------------------------------------------------------------------------------------
KAS_DEV uint32_t mid32_pack(int32_t h0, int32_t h1, int32_t h2);
KAS_DEV void mid32_fields(uint32_t w, uint32_t (&c)[3]);
// (m32, dword mid rows — KAS_FLAG_MID32: the row's holders are kept in registers from the dword the row scan stored, and every
// accept stores the whole row again, sorted)
template <int W>
KAS_DEV int32_t p4_window(const LdsView& L, int32_t count, int32_t cap, int32_t live_count,
                          int32_t& head, uint16_t* mid, int32_t mw, int64_t (&st)[8], bool m32 = false) {
  const int lane = kasw::lane();
  const bool mine = lane < count;
  const int32_t p = mine ? L.ring_p[lane] : 0;
  int32_t cells[W];
#pragma unroll
  for (int k = 0; k < W; ++k) cells[k] = -1;
  if constexpr (W == 3) if (m32 && mine) {
    uint32_t f[3];
    mid32_fields(reinterpret_cast<const uint32_t*>(mid)[p], f);
#pragma unroll
    for (int k = 0; k < 3; ++k) cells[k] = f[k] == 0x7ffu ? -1 : (int32_t)f[k];
  }
  const int32_t meta = mine ? L.ring_meta[lane] : 0;
  int32_t need = meta & 0xff;
  int32_t hc = (meta >> 8) & 0xff;
  int32_t hr[W];
#pragma unroll
  for (int j = 0; j < W; ++j) hr[j] = mine ? (int32_t)L.ring_rack[j * KAS_RING_CAP + lane] : -1;
  kasw::lockstep();   // ring fully read before the caller refills it

  int32_t fail_lane = -1;
  int32_t j = head;
  KAS_COUNT(st[4]);
  constexpr int U = 4;                                     // node positions fetched per LDS round trip
  for (;;) {
    uint64_t pend = kasw::ballot(need > 0);
    if (pend == 0) break;
    if (j >= live_count) { fail_lane = kasw::first_lane(pend); break; }
    // the next U non-full nodes in processing order: their table entries in one go (each node
    // is listed once, and only this wave changes load[], so the values stay valid below)
    int32_t n[U], slots[U], rk[U];
#pragma unroll
    for (int u = 0; u < U; ++u) n[u] = (int32_t)L.live[j + u < live_count ? j + u : j];
#pragma unroll
    for (int u = 0; u < U; ++u) { slots[u] = cap - lds_load(L, n[u]); rk[u] = (int32_t)lds_rack(L, n[u]); }
    int32_t taken[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      taken[u] = 0;
      if (j + u < live_count && pend != 0) {               // wave-uniform
        KAS_COUNT(st[5]);
        if (slots[u] > 0) {
          bool want = need > 0;
#pragma unroll
          for (int k = 0; k < W; ++k) want = want && !(k < hc && hr[k] == rk[u]);
          const uint64_t w = kasw::ballot(want);
          if (w != 0) {
            const int32_t rank = kasw::count_below(w);
            if (want && rank < slots[u]) {                 // accept (KAS:178-181)
              if (W == 3 && m32) {
                put<W>(cells, hc, n[u]);
                if constexpr (W == 3) reinterpret_cast<uint32_t*>(mid)[p] = mid32_pack(cells[0], cells[1], cells[2]);
              } else {
                mid[(int64_t)p * mw + hc] = (uint16_t)n[u];
              }
              put<W>(hr, hc, rk[u]);
              hc += 1;
              need -= 1;
            }
            const int32_t takers = kasw::popc(w);
            taken[u] = takers < slots[u] ? takers : slots[u];
            pend = kasw::ballot(need > 0);
          }
        }
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) if (taken[u] > 0) lds_load(L, n[u]) += taken[u];
    }
    kasw::lockstep();
    j += U;
  }
  // drop the leading nodes that are now full from future windows
  while (head < live_count && lds_load(L, (int32_t)L.live[head]) >= cap) ++head;
  return fail_lane;
}

// Intermediate ("mid") rows.  Between the fill kernel and the order kernel a row is the node indices
// of its holders in acceptance order as uint16 (KAS_MID_NONE = no holder; N <= 32767 is a plan
// limit), mid_width(ow) of them per row, stored at the END of the topic's out region: half the
// bytes of an int32 row for the fill to write and the order kernel to read.  The order kernel turns
// rows into final out rows (broker ids, int32) in ascending row order from the START of the same
// region: final row p ends at 4 ow (p + 1) bytes, mid row q starts at 4 ow P - 2 mw (P - q), and
// 2 mw <= 4 ow, so a final row never reaches a mid row that is still to be read (q > p).
// Round 5: rows are PACKED, mid_width(ow) == ow — 6 bytes at lists 3 wide where rounds 2-4 padded to 8 (and 10 for 12
// at lists 5 wide): a quarter of the intermediate bytes, which the fill kernel's rate follows (alone on the GPU it ran
// 28 % faster in a tuning build without its row stores).  A row is then only 2-byte aligned: it is moved as dwords plus
// one trailing halfword through 2-byte-aligned accesses (U32a2: one global_load_dword / global_store_dword each — the
// hardware's unaligned access mode, which the HSA ABI turns on).
#define KAS_MID_NONE 0xffffu
#ifndef KAS_IXROWS_SKIP_SAME
#define KAS_IXROWS_SKIP_SAME 1
#endif
#ifndef KAS_MID_PAD
#define KAS_MID_PAD 0                    // tuning builds: 1 = rows padded to an even number of cells (rounds 2-4), same accesses
#endif
KAS_DEV int32_t mid_width(int32_t ow) { return KAS_MID_PAD ? ((ow + 1) & ~1) : ow; }
template <int W>
constexpr int mid_width_of() { return KAS_MID_PAD ? ((W + 1) & ~1) : W; }
struct __attribute__((packed, aligned(2))) U32a2 { uint32_t v; };
KAS_DEV uint32_t load_u32_a2(const uint16_t* p) { return reinterpret_cast<const U32a2*>(p)->v; }
KAS_DEV void store_u32_a2(uint16_t* p, uint32_t v) { reinterpret_cast<U32a2*>(p)->v = v; }
KAS_DEV uint16_t* mid_base(int32_t* out, int32_t P, int32_t ow) {
  return (uint16_t*)(out + (int64_t)P * ow) - (int64_t)P * mid_width(ow);
}
KAS_DEV int32_t mid_to_index(uint32_t v) { return (v & 0x8000u) ? -1 : (int32_t)v; }   // (bit 15: KAS_MID_NONE — a node index is below 32768)

// Dword mid rows (KAS_FLAG_MID32, round 6).  Nothing downstream of the fill depends on the order of a row's holders: first fit
// appends to them, and the preference ordering visits them in ascending node order (KAS:228's TreeSet, KAS:188-200).  A row of
// up to three holders is therefore kept SORTED, a <= b <= c with "no holder" (0x7ff) last, and three 11-bit fields need only 32
// bits because the top bits of a sorted triple are monotone: the middle value b is stored whole (bits 0..10); if its top bit is
// set, so is c's (c keeps 10 bits, a all 11), if it is clear, so is a's (a keeps 10 bits, c all 11).  One aligned dword per row
// whatever the topic's width: 4 bytes for the fill to write and the order kernel to read where the packed uint16 row is 6, and
// one memory instruction where that one takes two.  Rows live at the END of the topic's out region like the 16-bit ones
// (mid row q starts at 4 ow P - 4 (P - q); final row p ends at 4 ow (p + 1) <= that for every q > p).
#define KAS_M32_NONE 0x7ffu
KAS_DEV bool mid32(const KasLaunch& a) { return (a.flags & KAS_FLAG_MID32) != 0u; }
// holders in any order, -1 (any value >= 0x7ff as unsigned) = none
KAS_DEV uint32_t mid32_pack(int32_t h0, int32_t h1, int32_t h2) {
  const uint32_t u0 = (uint32_t)h0, u1 = (uint32_t)h1, u2 = (uint32_t)h2;
  const uint32_t mn01 = u0 < u1 ? u0 : u1, mx01 = u0 < u1 ? u1 : u0;
  const uint32_t lo = mn01 < u2 ? mn01 : u2;
  const uint32_t hi = mx01 < u2 ? u2 : mx01;
  const uint32_t md0 = mn01 < u2 ? u2 : mn01;               // max(min(u0, u1), u2) ...
  const uint32_t md = md0 < mx01 ? md0 : mx01;              // ... capped by max(u0, u1): the median
  const uint32_t a = lo < KAS_M32_NONE ? lo : KAS_M32_NONE, b = md < KAS_M32_NONE ? md : KAS_M32_NONE, c = hi < KAS_M32_NONE ? hi : KAS_M32_NONE;
  const bool f = (b & 0x400u) != 0u;
  const uint32_t x = f ? a : c, y = f ? (c & 0x3ffu) : a;
  return b | (x << 11) | (y << 22);
}
// ... and back: ascending node indices, KAS_M32_NONE = none (last)
KAS_DEV void mid32_fields(uint32_t w, uint32_t (&c)[3]) {
  const uint32_t b = w & 0x7ffu, x = (w >> 11) & 0x7ffu, y = w >> 22;
  const bool f = (w & 0x400u) != 0u;
  c[0] = f ? x : y; c[1] = b; c[2] = f ? (y | 0x400u) : x;
}
// the same row in the layout of the packed 16-bit rows (MidRaw<3>: cells 0, 1 in w[0], cell 2 in w[1]; KAS_MID_NONE = none)
KAS_DEV void mid32_to_raw16(uint32_t w, uint32_t& w0, uint32_t& w1) {
  uint32_t c[3];
  mid32_fields(w, c);
#pragma unroll
  for (int k = 0; k < 3; ++k) c[k] = c[k] == KAS_M32_NONE ? KAS_MID_NONE : c[k];
  w0 = c[0] | (c[1] << 16); w1 = c[2] | 0xffff0000u;
}

// ---- cells of the cur / out tables: int32 broker ids, or — KAS_FLAG_CELLS16, the plans of kas_plan_create16 (ABI v5) —
// uint16 node indices (0xFFFF: no such broker / pad; node i has id i).  With 16-bit cells a topic's out region is exactly
// its mid rows' size (ow cells of 2 bytes a row): final row p takes the place of mid row p, which every order kernel has
// read before it writes the row.
KAS_DEV bool cells16(const KasLaunch& a) { return (a.flags & KAS_FLAG_CELLS16) != 0u; }
struct OutRef {
  int32_t* o32;          // int32 cells (nullptr with 16-bit cells)
  uint16_t* o16;         // uint16 cells
};
KAS_DEV OutRef topic_out(const KasLaunch& a, const kas_topic_desc& td) {
  OutRef o;
  o.o32 = cells16(a) ? nullptr : a.out + td.out_off;
  o.o16 = cells16(a) ? reinterpret_cast<uint16_t*>(a.out) + td.out_off : nullptr;
  return o;
}
KAS_DEV void out_store(const OutRef& o, int64_t cell, int32_t v) {   // v: broker id / node index, or -1 = pad
  if (o.o16) o.o16[cell] = (uint16_t)v;                               // (-1 -> 0xFFFF)
  else o.o32[cell] = v;
}
KAS_DEV void out_pad(const OutRef& o, int64_t cells, int32_t first, int32_t step) {   // a topic that returns nothing: all padding
  if (o.o16) { for (int64_t i = first; i < cells; i += step) o.o16[i] = (uint16_t)0xffffu; }
  else { for (int64_t i = first; i < cells; i += step) o.o32[i] = -1; }
}
KAS_DEV uint16_t* topic_mid(const KasLaunch& a, const kas_topic_desc& td) {
  if (cells16(a)) return reinterpret_cast<uint16_t*>(a.out) + td.out_off;
  if (mid32(a))                                              // one dword per row, at the end of the topic's out region
    return reinterpret_cast<uint16_t*>(a.out + td.out_off + (int64_t)td.n_partitions * td.out_width - td.n_partitions);
  return mid_base(a.out + td.out_off, td.n_partitions, td.out_width);
}
// the topic's cur rows: for 16-bit cells the pointer is only a byte address (2-byte aligned), read through load_row / cur_cell
KAS_DEV const int32_t* topic_cur(const KasLaunch& a, const kas_topic_desc& td) {
  if (cells16(a)) return reinterpret_cast<const int32_t*>(reinterpret_cast<const uint16_t*>(a.cur) + td.cur_off);
  return a.cur + td.cur_off;
}

// One mid row, as loaded: nothing is computed from the loaded values here, so a row read ahead of its
// use (every consumer prefetches) does not make the wave wait for the load where it is issued.
// Rows of the template width (ow == W) come as W / 2 dwords and, W odd, one halfword (w[k] = cells 2 k, 2 k + 1; the
// cell a row does not have reads KAS_MID_NONE), others cell by cell.
template <int W>
struct MidRaw { uint32_t w[W]; };
// (m32: dword mid rows, KAS_FLAG_MID32 — the row's one dword in w[0])
template <int W>
KAS_DEV MidRaw<W> mid_load_raw(const uint16_t* mid, int32_t ow, int64_t p, bool active, bool m32 = false) {
  MidRaw<W> raw;
#pragma unroll
  for (int k = 0; k < W; ++k) raw.w[k] = 0xffffffffu;
  if (W == 3 && m32) {
    if (active) raw.w[0] = reinterpret_cast<const uint32_t*>(mid)[p];
  } else if (ow == W) {
    if (active) {
      const uint16_t* row = mid + p * mid_width_of<W>();
#pragma unroll
      for (int k = 0; k < W / 2; ++k) raw.w[k] = load_u32_a2(row + 2 * k);
      if constexpr (W & 1) raw.w[W / 2] = (uint32_t)row[W - 1] | 0xffff0000u;
    }
  } else {
    const int32_t mw = mid_width(ow);
#pragma unroll
    for (int k = 0; k < W; ++k) if (active && k < ow) raw.w[k] = (uint32_t)mid[p * mw + k];
  }
  return raw;
}
// ... and unpacked where it is used: node indices, -1 = none
template <int W>
KAS_DEV void mid_unpack(const MidRaw<W>& raw, int32_t ow, int32_t (&c)[W], bool m32 = false) {
  if constexpr (W == 3) if (m32) {
    uint32_t f[3];
    mid32_fields(raw.w[0], f);
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = f[k] == KAS_M32_NONE ? -1 : (int32_t)f[k];
    return;
  }
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const uint32_t packed = (raw.w[k >> 1] >> (16 * (k & 1))) & 0xffffu;
    const uint32_t v = ow == W ? packed : (raw.w[k] & 0xffffu);
    c[k] = mid_to_index(v);
  }
}

// Everything a phase needs to know about the topic being solved.
struct TopicView {
  bool c16;             // cells of cur are uint16 node indices (KAS_FLAG_CELLS16)
  bool m32 = false;     // mid rows are one dword each (KAS_FLAG_MID32): `mid` points at row 0's dword
  const int32_t* cur;
  uint16_t* mid;        // mid rows of the topic (at the end of its out region); mid_width(ow) uint16 each
  int32_t* orph;        // orphan row lists of this scenario (HBM scratch)
  const int32_t* len_arr;
  const int32_t* inp_arr;
  const int32_t* pid_arr;
  int32_t P, cw, rf, ow, hash, nt, N, cap;
};

// current replica list of row p (ids beyond cur_width read as -1; validity comes from len).
// FULL: every list of the topic is exactly W wide (cur_width == W, no cur_len array) — the whole
// row is one W-dword load and needs no per-cell tests; rows past the end re-read the last row
// with len = 0.
template <int W>
struct RowW { int32_t v[W]; };

// C16: the cells are uint16 node indices (a full row: the packed 2 W bytes of a mid row, read the same way)
KAS_DEV int32_t cur_cell(const TopicView& T, int64_t cell) {
  if (T.c16) return mid_to_index((uint32_t)reinterpret_cast<const uint16_t*>(T.cur)[cell]);
  return T.cur[cell];
}
template <int W, bool FULL = false, bool C16 = false>
KAS_DEV void load_row(const TopicView& T, int32_t p, int32_t (&ids)[W], int32_t& len) {
  const bool active = p < T.P;
  if constexpr (FULL) {
    const int32_t pc = active ? p : (T.P > 0 ? T.P - 1 : 0);
    if constexpr (C16) {
      static_assert(!KAS_MID_PAD, "16-bit cells: rows are packed");
      const MidRaw<W> q = mid_load_raw<W>(reinterpret_cast<const uint16_t*>(T.cur), W, pc, true);
      mid_unpack<W>(q, W, ids);
    } else {
      const RowW<W> q = *reinterpret_cast<const RowW<W>*>(T.cur + (int64_t)pc * W);
#pragma unroll
      for (int r = 0; r < W; ++r) ids[r] = q.v[r];
    }
    len = active ? W : 0;
  } else {
    if constexpr (C16) {
      const uint16_t* c = reinterpret_cast<const uint16_t*>(T.cur);
#pragma unroll
      for (int r = 0; r < W; ++r) ids[r] = (active && r < T.cw) ? mid_to_index((uint32_t)c[(int64_t)p * T.cw + r]) : -1;
    } else {
#pragma unroll
      for (int r = 0; r < W; ++r) ids[r] = (active && r < T.cw) ? T.cur[(int64_t)p * T.cw + r] : -1;
    }
    len = active ? (T.len_arr ? T.len_arr[p] : T.cw) : 0;
  }
}

// every current list of the topic is exactly W wide, and every out row too
template <int W>
KAS_DEV bool full_rows_of(const TopicView& T) { return T.cw == W && T.ow == W && T.len_arr == nullptr; }

// Stream the tiles tile0, tile0 + stride, ... (< t_end) of the cur table through `body(tile, ids,
// len)` with KAS_TILES_AHEAD tiles of rows in flight per lane: the scans are HBM-latency-bound
// (12 bytes per lane per tile), so several tiles are requested before the first is consumed.
#ifndef KAS_TILES_AHEAD
#define KAS_TILES_AHEAD 4
#endif
template <int W, bool FULL, bool C16, typename Body>
KAS_DEV void for_tiles_impl(const TopicView& T, int32_t tile0, int32_t stride, int32_t t_end, Body body) {
  constexpr int D = KAS_TILES_AHEAD;
  const int lane = kasw::lane();
  int32_t nx[D][W], nlen[D];
#pragma unroll
  for (int d = 0; d < D; ++d) load_row<W, FULL, C16>(T, ((tile0 + d * stride) << 6) + lane, nx[d], nlen[d]);
  for (int32_t tile = tile0; tile < t_end; tile += stride * D) {
    int32_t ids[D][W], len[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int r = 0; r < W; ++r) ids[d][r] = nx[d][r];
      len[d] = nlen[d];
    }
#pragma unroll
    for (int d = 0; d < D; ++d)                              // request the next batch
      load_row<W, FULL, C16>(T, ((tile + (D + d) * stride) << 6) + lane, nx[d], nlen[d]);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int32_t t = tile + d * stride;
      if (t < t_end) body(t, ids[d], len[d]);                // wave-uniform
    }
  }
}

// Same stream, but `body(ids[D][W], len[D])` gets KAS_TILES_AHEAD tiles at once (len = 0 for
// rows / tiles past the end): for passes whose tiles do not depend on each other, so that the
// LDS lookups of the four tiles overlap instead of forming four serial chains.
template <int W, bool FULL, bool C16, typename Body>
KAS_DEV void for_tile_batches_impl(const TopicView& T, int32_t tile0, int32_t stride, int32_t t_end, Body body) {
  constexpr int D = KAS_TILES_AHEAD;
  const int lane = kasw::lane();
  int32_t nx[D][W], nlen[D];
#pragma unroll
  for (int d = 0; d < D; ++d) load_row<W, FULL, C16>(T, ((tile0 + d * stride) << 6) + lane, nx[d], nlen[d]);
  for (int32_t tile = tile0; tile < t_end; tile += stride * D) {
    int32_t ids[D][W], len[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int r = 0; r < W; ++r) ids[d][r] = nx[d][r];
      len[d] = (tile + d * stride < t_end) ? nlen[d] : 0;
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
      load_row<W, FULL, C16>(T, ((tile + (D + d) * stride) << 6) + lane, nx[d], nlen[d]);
    body(ids, len);
  }
}

template <int W>
KAS_DEV bool full_rows(const TopicView& T) { return full_rows_of<W>(T); }

template <int W, typename Body>
KAS_DEV void for_tiles(const TopicView& T, int32_t tile0, int32_t stride, int32_t t_end, Body body) {
  if constexpr (W <= 3) if (T.c16) {                         // (16-bit cells: lists up to 3 wide; wave-uniform: one row stream runs)
    if (full_rows<W>(T)) for_tiles_impl<W, true, true>(T, tile0, stride, t_end, body);
    else for_tiles_impl<W, false, true>(T, tile0, stride, t_end, body);
    return;
  }
  {
    if (full_rows<W>(T)) for_tiles_impl<W, true, false>(T, tile0, stride, t_end, body);
    else for_tiles_impl<W, false, false>(T, tile0, stride, t_end, body);
  }
}

template <int W, typename Body>
KAS_DEV void for_tile_batches(const TopicView& T, int32_t tile0, int32_t stride, int32_t t_end, Body body) {
  if constexpr (W <= 3) if (T.c16) {
    if (full_rows<W>(T)) for_tile_batches_impl<W, true, true>(T, tile0, stride, t_end, body);
    else for_tile_batches_impl<W, false, true>(T, tile0, stride, t_end, body);
    return;
  }
  {
    if (full_rows<W>(T)) for_tile_batches_impl<W, true, false>(T, tile0, stride, t_end, body);
    else for_tile_batches_impl<W, false, false>(T, tile0, stride, t_end, body);
  }
}

// ---------------------------------------------------------------------------------------------
// P3 for one row per lane (KAS:133-160): holders of the row from its accepted replicas, orphan
// count, movement bookkeeping and the out row (node indices for now).
// ---------------------------------------------------------------------------------------------
template <int W>
KAS_DEV void p3_rows(const LdsView& L, const TopicView& T, int32_t p, int32_t len,
                     const int32_t (&ids)[W], const int32_t (&idx)[W], uint32_t accbits,
                     int32_t& need, int32_t& hc, int32_t (&hrack)[W], int32_t& moved_r,
                     int32_t& moved_p, const int32_t* racks = nullptr, bool inplace = false) {
  const bool active = p < T.P;
  int32_t hold[W];
#pragma unroll
  for (int k = 0; k < W; ++k) { hold[k] = -1; hrack[k] = -1; }
  hc = 0;
#pragma unroll
  for (int r = 0; r < W; ++r) {
    const bool acc = (accbits >> r) & 1u;
    const int32_t n = idx[r] >= 0 ? idx[r] : 0;
    const int32_t rk = racks ? racks[r] : (int32_t)lds_rack(L, n);   // the caller may have looked it up already
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const bool here = acc && hc == k;
      hold[k] = here ? n : hold[k];
      hrack[k] = here ? rk : hrack[k];
    }
    hc += acc ? 1 : 0;
  }
  const bool in_parts = active && (T.inp_arr ? T.inp_arr[p] != 0 : true);
  need = in_parts ? (T.rf - hc > 0 ? T.rf - hc : 0) : 0;   // KAS:151-157
#if defined(KAS_TUNE_NO_ORPHANS)                            // (tuning builds, WRONG results: what would the job run at if P4 cost the fill nothing?
  {                                                         //  missing holders are filled with other brokers on the spot, no row is an orphan)
#pragma unroll
    for (int k = 0; k < W; ++k) hold[k] = (k >= hc && k < T.rf) ? (int32_t)((uint32_t)(p * 7 + k * 131) % (uint32_t)T.N) : hold[k];
    need = 0;
  }
#endif
  // inplace (index rows, fill_pass_a_fused<EMIT>): the row's cells already lie where its mid row goes — a row that keeps every
  // one of them, in order, is not stored again (KAS_IXROWS_SKIP_SAME=0: tuning builds store every row)
  bool same = inplace && KAS_IXROWS_SKIP_SAME;
#pragma unroll
  for (int k = 0; k < W; ++k) same = same && hold[k] == idx[k];
#if defined(KAS_TUNE_NO_MID_STORES)                          // (tuning builds: how much of the fill's time is its 8-byte row stores)
  if (false) {
#else
  if (active && !same) {
#endif
    if (W == 3 && T.m32) {                                  // wave-uniform: the row's holders sorted, in one dword
      if constexpr (W == 3) reinterpret_cast<uint32_t*>(T.mid)[p] = mid32_pack(hold[0], hold[1], hold[2]);
    } else if (T.ow == W) {                                 // wave-uniform: the mid row as W / 2 dwords (+ a halfword)
      uint16_t* row = T.mid + (int64_t)p * mid_width_of<W>();
#pragma unroll
      for (int k = 0; k < W / 2; ++k)
        store_u32_a2(row + 2 * k, ((uint32_t)hold[2 * k] & 0xffffu) | (((uint32_t)hold[2 * k + 1] & 0xffffu) << 16));
      if constexpr (W & 1) row[W - 1] = (uint16_t)hold[W - 1];
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k)
        if (k < mid_width(T.ow)) T.mid[(int64_t)p * mid_width(T.ow) + k] = (uint16_t)hold[k];
    }
  }
  // a distinct current broker that was not kept => set(new) != set(cur)
  uint32_t kept_else = 0;
#pragma unroll
  for (int r = 0; r < W; ++r)
#pragma unroll
    for (int r2 = 0; r2 < W; ++r2)
      kept_else |= (((accbits >> r2) & 1u) != 0u && ids[r2] == ids[r]) ? (1u << r) : 0u;
  const uint32_t present = len >= W ? ((1u << W) - 1u) : ((1u << len) - 1u);
  const bool dropped = (present & ~accbits & ~kept_else) != 0u;
  moved_r += need;
  moved_p += (active && (dropped || need > 0)) ? 1 : 0;
}

// append this tile's orphans (ascending lane == ascending row) to the LDS ring
template <int W>
KAS_DEV void ring_push(const LdsView& L, int32_t p, int32_t need, int32_t hc,
                       const int32_t (&hrack)[W], int32_t& ring_count) {
  const bool orphan = need > 0;
  const uint64_t om = kasw::ballot(orphan);
  if (orphan) {
    const int32_t slot = ring_count + kasw::count_below(om);
    L.ring_p[slot] = p;
    L.ring_meta[slot] = need | (hc << 8);
#pragma unroll
    for (int k = 0; k < W; ++k) L.ring_rack[k * KAS_RING_CAP + slot] = (int16_t)hrack[k];
  }
  ring_count += kasw::popc(om);
  kasw::lockstep();
}

// Run P4 windows while the ring holds at least `min_fill` orphans (64 inside the row scan, 1 at
// its end).  Returns the failing row (KAS:183-184) or -1.
template <int W>
KAS_DEV int32_t drain_ring(const LdsView& L, const TopicView& T, int32_t min_fill, int32_t live_count,
                           int32_t& head, int32_t& ring_count, int64_t (&st)[8]) {
  const int lane = kasw::lane();
  while (ring_count >= min_fill && ring_count > 0) {
    const int32_t n_win = ring_count < 64 ? ring_count : 64;
    const int32_t fl = p4_window<W>(L, n_win, T.cap, live_count, head, T.mid, mid_width(T.ow), st, T.m32);
    if (fl >= 0) return L.ring_p[fl];
    const int32_t rest = ring_count - n_win;           // shift the ring down by one window
    int32_t tp = 0, tm = 0; int32_t tr[W];
#pragma unroll
    for (int k = 0; k < W; ++k) tr[k] = 0;
    if (lane < rest) {
      tp = L.ring_p[64 + lane]; tm = L.ring_meta[64 + lane];
#pragma unroll
      for (int k = 0; k < W; ++k) tr[k] = L.ring_rack[k * KAS_RING_CAP + 64 + lane];
    }
    kasw::lockstep();
    if (lane < rest) {
      L.ring_p[lane] = tp; L.ring_meta[lane] = tm;
#pragma unroll
      for (int k = 0; k < W; ++k) L.ring_rack[k * KAS_RING_CAP + lane] = (int16_t)tr[k];
    }
    ring_count = rest;
    kasw::lockstep();
  }
  return -1;
}

// ---------------------------------------------------------------------------------------------
// P2, general form (KAS:101-131): one sweep per replica index, accept-mask words in HBM.
// Executed by ONE wave (the other waves of the workgroup wait at the next barrier).
// ---------------------------------------------------------------------------------------------
template <int W>
KAS_DEV_COLD void fill_generic_sweeps(const LdsView& L, const TopicView& T, const NodeMap& nm,
                                 uint64_t* accmask, int64_t (&st)[8]) {
  const int lane = kasw::lane();
  const int32_t cap = T.cap, nt = T.nt;
  if (T.P <= 0 || T.cw <= 0) return;                         // (no rows: nothing to sweep, and no row for the row stream to read)
  // (round 6: the rows come through the row stream of the other scans — KAS_TILES_AHEAD tiles asked for before the first is used —
  //  instead of a load per cell at the point of use: this wavefront is alone on its chain, and a tile then costs its LDS round
  //  trips, not an HBM round trip per sweep as well; a batch of 1000 scenarios that all take this form: 20.2 -> 16.1 ms — the rest is
  //  this one wavefront's chain of LDS round trips per tile and first fit from the ring)
  for (int32_t r = 0; r < T.cw; ++r) {
    for_tiles<W>(T, 0, 1, nt, [&](int32_t tile, const int32_t (&ids)[W], int32_t len) {
      int32_t n = -1;
      if (r < len) n = node_lookup(L, nm, sel<W>(ids, r));  // node != null (KAS:119-120)
      bool elig = n >= 0;
      const int32_t rk = elig ? (int32_t)lds_rack(L, n) : -1;
#pragma unroll
      for (int r2 = 0; r2 < W - 1; ++r2) {
        if (r2 < r) {                                       // wave-uniform
          const uint64_t aw = kasw::load_shared_u64(accmask + (int64_t)r2 * nt + tile);
          if (elig && ((aw >> lane) & 1ull)) {
            const int32_t n2 = node_lookup(L, nm, ids[r2]);
            if ((int32_t)lds_rack(L, n2) == rk) elig = false;    // rack.canAccept (KAS:346-348)
          }
        }
      }
      const bool took = elig && lds_load(L, n) < cap;            // size() < capacity (KAS:322)
      kasw::lockstep();                                     // every lane saw the pre-tile load
      if (took) kasw::lds_atomic_add(&lds_load(L, n), 1);
      kasw::lockstep();
      bool accepted = took;
      uint64_t todo = kasw::ballot(took && lds_load(L, n) > cap);
      if (todo != 0) {
        KAS_COUNT(st[7]);
        // some node overflowed inside this tile: keep its first (cap - load_before) lanes
        while (todo != 0) {
          const int leader = kasw::first_lane(todo);
          const int32_t t = kasw::shfl(n, leader);
          const bool same_l = took && n == t;
          const uint64_t same = kasw::ballot(same_l);
          const int32_t before = lds_load(L, t) - kasw::popc(same);
          if (same_l) accepted = before + kasw::count_below(same) < cap;
          kasw::lockstep();                                 // all lanes read load[t]
          if (lane == leader) lds_load(L, t) = cap;
          todo &= ~same;
        }
        kasw::lockstep();
      }
      const uint64_t accw = kasw::ballot(accepted);
      if (lane == 0) kasw::store_shared_u64(accmask + (int64_t)r * nt + tile, accw);
    });
    kasw::wave_sync();   // this sweep's mask words are visible to the next sweep's loads
  }
}

// P3 + P4 over the accept-mask words of the general sticky fill (one wave)
template <int W>
KAS_DEV_COLD int32_t p3p4_generic(const LdsView& L, const TopicView& T, const NodeMap& nm,
                             const uint64_t* accmask, int32_t live_count, int32_t& moved_r,
                             int32_t& moved_p, int64_t (&st)[8]) {
  const int lane = kasw::lane();
  int32_t ring_count = 0, head = 0;
  int32_t failed = -1;                                       // (wave-uniform: the row first fit could not place)
  if (T.P > 0)                                               // (the row stream, as in fill_generic_sweeps)
  for_tiles<W>(T, 0, 1, T.nt, [&](int32_t tile, const int32_t (&ids)[W], int32_t len) {
    if (failed >= 0) return;
    const int32_t p = (tile << 6) + lane;
    int32_t idx[W];
    uint32_t accbits = 0;
#pragma unroll
    for (int r = 0; r < W; ++r) {
      idx[r] = -1;
      if (r < T.cw) {                                       // wave-uniform
        const uint64_t aw = kasw::load_shared_u64(accmask + (int64_t)r * T.nt + tile);
        if (p < T.P && ((aw >> lane) & 1ull)) {
          accbits |= 1u << r;
          idx[r] = node_lookup(L, nm, ids[r]);
        }
      }
    }
    int32_t need, hc, hrack[W];
    p3_rows<W>(L, T, p, len, ids, idx, accbits, need, hc, hrack, moved_r, moved_p);
    ring_push<W>(L, p, need, hc, hrack, ring_count);
    const int32_t fr = drain_ring<W>(L, T, 64, live_count, head, ring_count, st);
    if (fr >= 0) failed = fr;
  });
  if (failed >= 0) return failed;
  return drain_ring<W>(L, T, 1, live_count, head, ring_count, st);
}

// ---------------------------------------------------------------------------------------------
// P2, rack-diverse form: passes A1, Q, A2, prefix, B (see the header comment).
// ---------------------------------------------------------------------------------------------
// A1: tiles wave, wave+NW, ... ; returns this lane's "not rack-diverse" verdict
// H16: hist[W][n] as uint16 cells, two to a dword (fewer than 65,536 rows in the range: the spread fill's chunks)
template <int W, bool DIRECT, bool H16 = false>
KAS_DEV bool fill_pass_a_range(const LdsView& L, const TopicView& T, const NodeMap& nm, int32_t tile0,
                               int32_t stride, int32_t t_end) {
  constexpr int D = KAS_TILES_AHEAD;
  const int32_t N = T.N;
  bool viol = false;
  for_tile_batches<W>(T, tile0, stride, t_end, [&](const int32_t (&ids)[D][W], const int32_t (&len)[D]) {
    int32_t idx[D][W], rk[D][W];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int r = 0; r < W; ++r) idx[d][r] = r < len[d] ? node_lookup_as<DIRECT>(L, nm, ids[d][r]) : -1;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int r = 0; r < W; ++r)
        rk[d][r] = idx[d][r] >= 0 ? (int32_t)lds_rack(L, idx[d][r]) : -1 - r;   // invalid: never equal
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int r = 1; r < W; ++r)
#pragma unroll
        for (int r2 = 0; r2 < r; ++r2) viol = viol || rk[d][r] == rk[d][r2];
#pragma unroll
      for (int r = 0; r < W; ++r) {
        if (idx[d][r] >= 0) {
          const int32_t cell = r * N + idx[d][r];
          if (H16) kasw::lds_atomic_add(&L.x[cell >> 1], 1 << (16 * (cell & 1)));
          else kasw::lds_atomic_add(&L.x[cell], 1);
        }
      }
    }
  });
  return viol;
}
template <int W, int NW, bool DIRECT>
KAS_DEV bool fill_pass_a(const LdsView& L, const TopicView& T, const NodeMap& nm, int32_t wave) {
  return fill_pass_a_range<W, DIRECT>(L, T, nm, wave, NW, T.nt);
}

template <int W, int NW>
constexpr int fused_block_words() {                                     // == kas_fused_block_words
  return ((((NW * W + 1) / 2) > NW + 2 ? ((NW * W + 1) / 2) : NW + 2) + 1) | 1;
}

// A1, fused form: wave w scans chunk w (the rows pass B will walk) and counts per chunk: x is node-major,
// kas_fused_block_words() dwords per node holding uint16 hist[n][w][r] (a chunk has < 65536 rows: the
// plan checks), so that the quota pass below also knows every chunk's share and no second counting
// pass over cur is needed.  Returns this lane's "not rack-diverse" verdict.
// IDENT: 16-bit cells (KAS_FLAG_CELLS16) — a cell IS its node index (node i has id i), no table to look into
KAS_DEV int32_t node_lookup_ident(const NodeMap& m, int32_t id) {
  const uint32_t d = (uint32_t)id - (uint32_t)m.min_id;
  return d < m.range ? (int32_t)d : -1;
}
// EMIT (KAS_FLAG_INDEX_ROWS; int32 cells, rows exactly W wide): the pass also leaves every row's node indices — the 2 W packed
// bytes of a mid row, 0xffff for a broker that is not in the set — where the topic's mid rows go (the end of its out region,
// scratch until pass B writes there).  Pass B then streams THOSE rows (6 bytes instead of 12 at lists 3 wide, no id lookups,
// and a row whose holders are its own cells — most rows — is already in place): `cur` is read once.
template <int W, int NW, bool DIRECT, bool IDENT = false, bool EMIT = false>
KAS_DEV bool fill_pass_a_fused(const LdsView& L, const TopicView& T, const NodeMap& nm, int32_t wave) {
  constexpr int D = KAS_TILES_AHEAD;
  constexpr int BW = fused_block_words<W, NW>();
  const int32_t t0 = chunk_begin<NW>(T.nt, wave), t1 = chunk_begin<NW>(T.nt, wave + 1);
  bool viol = false;
  int32_t tb = t0;                                             // first tile of the batch the body is handed (EMIT)
  const int lane = kasw::lane();
  for_tile_batches<W>(T, t0, 1, t1, [&](const int32_t (&ids)[D][W], const int32_t (&len)[D]) {
    int32_t idx[D][W], rk[D][W];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int r = 0; r < W; ++r) idx[d][r] = r < len[d] ? (IDENT ? node_lookup_ident(nm, ids[d][r]) : node_lookup_as<DIRECT>(L, nm, ids[d][r])) : -1;
    if constexpr (EMIT) {
      static_assert(W == 2 || W == 3, "index rows: lists 2 and 3 wide");
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (len[d] != 0) {                                     // (the row exists; full rows: len == W)
          uint16_t* row = T.mid + (int64_t)(((tb + d) << 6) + lane) * W;
          store_u32_a2(row, ((uint32_t)idx[d][0] & 0xffffu) | ((uint32_t)idx[d][1] << 16));
          if constexpr (W == 3) row[2] = (uint16_t)idx[d][2];
        }
      }
      tb += D;
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int r = 0; r < W; ++r)
        rk[d][r] = idx[d][r] >= 0 ? (int32_t)lds_rack(L, idx[d][r]) : -1 - r;   // invalid: never equal
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
      for (int r = 1; r < W; ++r)
#pragma unroll
        for (int r2 = 0; r2 < r; ++r2) viol = viol || rk[d][r] == rk[d][r2];
#pragma unroll
      for (int r = 0; r < W; ++r) {
        const int32_t cell = wave * W + r;                   // uint16 cell of the node's block
        if (idx[d][r] >= 0) kasw::lds_atomic_add(&L.x[idx[d][r] * BW + (cell >> 1)], 1 << (16 * (cell & 1)));
      }
    }
  });
  return viol;
}

// Q + chunk prefix, fused form: per node the sweep totals over all chunks -> r*, quota, load as in
// fill_quota; then, from the per-chunk counts of sweep r*, the quota left when chunk w starts, written
// over the first NW dwords of the node's own block (all its counts were read first).
template <int W, int NW>
KAS_DEV void fill_quota_fused(const LdsView& L, const TopicView& T, int32_t tid) {
  const int32_t N = T.N;
  constexpr int BW = fused_block_words<W, NW>();
  constexpr int HW = (NW * W + 1) / 2;                       // words that hold the uint16 counts
  for (int32_t n = tid; n < N; n += 64 * NW) {
    uint32_t words[HW];
#pragma unroll
    for (int k = 0; k < HW; ++k) words[k] = (uint32_t)L.x[n * BW + k];
    int32_t h[NW][W], tot[W];
#pragma unroll
    for (int r = 0; r < W; ++r) tot[r] = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
      for (int r = 0; r < W; ++r) {
        const int cell = w * W + r;
        h[w][r] = (int32_t)((words[cell >> 1] >> (16 * (cell & 1))) & 0xffffu);
        tot[r] += h[w][r];
      }
    int32_t cum = 0, rs = W, q = 0;
#pragma unroll
    for (int r = 0; r < W; ++r) {
      const int32_t c = tot[r];
      const bool sat = rs == W && c > T.cap - cum;          // cum + c > cap, overflow-safe
      q = sat ? T.cap - cum : q;
      cum = rs == W ? (sat ? T.cap : cum + c) : cum;
      rs = sat ? r : rs;
    }
    lds_load(L, n) = cum;
    lds_qrs(L, n) = (int32_t)(((uint32_t)rs << 28) | (uint32_t)q);
    int32_t rem = q;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      int32_t c = 0;
#pragma unroll
      for (int r = 0; r < W; ++r) c = rs == r ? h[w][r] : c;
      L.x[n * BW + w] = rem;
      rem -= c;
    }
  }
}

// Q: load[n] <- replicas the node keeps, qrs[n] <- r* << 28 | quota in sweep r*; the chunk rows
// of x (which alias the histogram rows of the same node) are cleared, or set to the quota when
// there is only one chunk.
template <int W, int NW>
KAS_DEV void fill_quota(const LdsView& L, const TopicView& T, int32_t tid) {
  const int32_t N = T.N;
  for (int32_t n = tid; n < N; n += 64 * NW) {
    int32_t cum = 0, rs = W, q = 0;
#pragma unroll
    for (int r = 0; r < W; ++r) {
      const int32_t c = L.x[r * N + n];
      const bool sat = rs == W && c > T.cap - cum;          // cum + c > cap, overflow-safe
      q = sat ? T.cap - cum : q;
      cum = rs == W ? (sat ? T.cap : cum + c) : cum;
      rs = sat ? r : rs;
    }
    lds_load(L, n) = cum;
    lds_qrs(L, n) = (int32_t)(((uint32_t)rs << 28) | (uint32_t)q);
#pragma unroll
    for (int w = 0; w < NW; ++w) L.x[w * N + n] = NW == 1 ? q : 0;
  }
}

// A2: wave w counts the sweep-r* candidates of chunk w per node
template <int W, int NW, bool DIRECT>
KAS_DEV void fill_chunk_count(const LdsView& L, const TopicView& T, const NodeMap& nm, int32_t wave) {
  constexpr int D = KAS_TILES_AHEAD;
  const int32_t N = T.N;
  const int32_t t0 = chunk_begin<NW>(T.nt, wave), t1 = chunk_begin<NW>(T.nt, wave + 1);
  int32_t* qc = L.x + wave * N;
  for_tile_batches<W>(T, t0, 1, t1, [&](const int32_t (&ids)[D][W], const int32_t (&len)[D]) {
    int32_t idx[D][W], rs[D][W];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int r = 0; r < W; ++r) idx[d][r] = r < len[d] ? node_lookup_as<DIRECT>(L, nm, ids[d][r]) : -1;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int r = 0; r < W; ++r)
        rs[d][r] = idx[d][r] >= 0 ? (int32_t)((uint32_t)lds_qrs(L, idx[d][r]) >> 28) : -1;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int r = 0; r < W; ++r) if (rs[d][r] == r) kasw::lds_atomic_add(&qc[idx[d][r]], 1);
  });
}

// prefix over chunks: x[w][n] <- quota of node n still unused when chunk w starts
template <int NW>
KAS_DEV void fill_chunk_prefix(const LdsView& L, const TopicView& T, int32_t tid) {
  const int32_t N = T.N;
  for (int32_t n = tid; n < N; n += 64 * NW) {
    int32_t rem = lds_qrs(L, n) & 0x0fffffff;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int32_t c = L.x[w * N + n];
      L.x[w * N + n] = rem;
      rem -= c;
    }
  }
}

// B + P3 over chunk `wave`; orphans go to the chunk's list; returns the number of orphans
// (FUSED: the quota words are x[n * BW + wave] of the node-major layout, else x[wave * N + n])
// RTN: the quota is drawn with one LDS atomic-with-return per list position — a node only ever counts at its own
// position r*, so all its draws of a tile are lanes of ONE instruction, served in ascending lane = row order
// (kas_wave.h, lds_add_rtn_u32; the launcher sets KAS_FLAG_LANE_ORDER where kas_ctx_create's self-test saw that order):
// what comes back is the quota left for exactly this row, no read before, no read after, no ranking of a tile in
// which a quota runs out.
template <int W, bool DIRECT, int QS, bool RTN = false, bool INPLACE = false>
KAS_DEV int32_t fill_pass_b_range(const LdsView& L, const TopicView& T, const NodeMap& nm, int32_t t0, int32_t t1,
                                  int32_t* qc, int32_t& moved_r, int32_t& moved_p, int64_t (&st)[8]) {
  const int lane = kasw::lane();
  int32_t* olist = T.orph + ((int64_t)t0 << 6);
  int32_t ocount = 0;
  for_tiles<W>(T, t0, 1, t1, [&](int32_t tile, const int32_t (&ids)[W], int32_t len) {
    const int32_t p = (tile << 6) + lane;
    int32_t idx[W], nn[W], before[W];
    uint32_t sure = 0, counting = 0;
#pragma unroll
    for (int r = 0; r < W; ++r) {
      int32_t rs;
      if constexpr (RTN && DIRECT) {
        // (the id table was rewritten behind the quota pass: node index | r* << 14 in one 16-bit word, fill_topic)
        const uint32_t d = (uint32_t)ids[r] - (uint32_t)nm.min_id;
        const uint32_t raw = (r < len && d < nm.range) ? (uint32_t)(uint16_t)L.idmap[d] : 0xffffu;
        idx[r] = raw != 0xffffu ? (int32_t)(raw & 0x3fffu) : -1;
        rs = (int32_t)(raw >> 14);
      } else {
        idx[r] = r < len ? node_lookup_as<DIRECT>(L, nm, ids[r]) : -1;
        rs = (int32_t)((uint32_t)lds_qrs(L, idx[r] >= 0 ? idx[r] : 0) >> 28);
      }
      nn[r] = idx[r] >= 0 ? idx[r] : 0;
      sure |= (idx[r] >= 0 && r < rs) ? (1u << r) : 0u;
      counting |= (idx[r] >= 0 && r == rs) ? (1u << r) : 0u;
      if constexpr (!RTN) before[r] = qc[nn[r] * QS];         // quota left before this tile
    }
    uint32_t accbits = sure;
    if constexpr (RTN) {
      if (kasw::ballot(counting != 0u) != 0) {
#pragma unroll
        for (int r = 0; r < W; ++r) {
          kasw::lockstep();
          int32_t left = 0;                                    // (only the drawing lanes execute the add)
          if ((counting >> r) & 1u) left = (int32_t)kasw::lds_add_rtn_u32((uint32_t*)&qc[nn[r] * QS], 0xffffffffu);
          accbits |= (((counting >> r) & 1u) && left > 0) ? (1u << r) : 0u;
        }
        kasw::lockstep();
      }
    } else if (kasw::ballot(counting != 0u) != 0) {
      kasw::lockstep();                                     // every lane saw the pre-tile quota
#pragma unroll
      for (int r = 0; r < W; ++r) if ((counting >> r) & 1u) kasw::lds_atomic_add(&qc[nn[r] * QS], -1);
      kasw::lockstep();
      uint32_t contested = 0;
#pragma unroll
      for (int r = 0; r < W; ++r) {
        const int32_t after = qc[nn[r] * QS];
        const bool cnt = (counting >> r) & 1u;
        accbits |= (cnt && after >= 0) ? (1u << r) : 0u;
        contested |= (cnt && before[r] > 0 && after < 0) ? (1u << r) : 0u;
      }
      uint64_t todo = kasw::ballot(contested != 0u);
      if (todo != 0) {
        KAS_COUNT(st[7]);
        // the quota of some node runs out inside this tile: rank its lanes (row order)
        while (todo != 0) {
          const int leader = kasw::first_lane(todo);
          int32_t mine_t = -1;
#pragma unroll
          for (int r = W - 1; r >= 0; --r) mine_t = ((contested >> r) & 1u) ? nn[r] : mine_t;
          const int32_t t = kasw::shfl(mine_t, leader);
          uint32_t hit = 0;                                  // my replica counted on node t
#pragma unroll
          for (int r = 0; r < W; ++r) hit |= (((counting >> r) & 1u) && nn[r] == t) ? (1u << r) : 0u;
          const uint64_t same = kasw::ballot(hit != 0u);
          const int32_t rank = kasw::count_below(same);
#pragma unroll
          for (int r = 0; r < W; ++r)
            accbits |= (((hit >> r) & 1u) && rank < before[r]) ? (1u << r) : 0u;
          contested &= ~hit;
          todo = kasw::ballot(contested != 0u);
        }
      }
    }
    int32_t need, hc, hrack[W];
    p3_rows<W>(L, T, p, len, ids, idx, accbits, need, hc, hrack, moved_r, moved_p, nullptr, INPLACE);
    const uint64_t om = kasw::ballot(need > 0);
    if (need > 0) olist[ocount + kasw::count_below(om)] = p;
    ocount += kasw::popc(om);
  });
  return ocount;
}
template <int W, int NW, bool DIRECT, bool FUSED = false, bool RTN = false, bool INPLACE = false>
KAS_DEV int32_t fill_pass_b(const LdsView& L, const TopicView& T, const NodeMap& nm, int32_t wave,
                            int32_t& moved_r, int32_t& moved_p, int64_t (&st)[8]) {
  const int32_t t0 = chunk_begin<NW>(T.nt, wave), t1 = chunk_begin<NW>(T.nt, wave + 1);
  constexpr int QS = FUSED ? fused_block_words<W, NW>() : 1;     // stride of a node's quota word
  int32_t* qc = FUSED ? L.x + wave : L.x + wave * T.N;
  return fill_pass_b_range<W, DIRECT, QS, RTN, INPLACE>(L, T, nm, t0, t1, qc, moved_r, moved_p, st);
}


// ---------------------------------------------------------------------------------------------
// P4 of the rack-diverse fill (KAS:162-186) on ALL wavefronts of the workgroup.  The orphan rows
// were listed per chunk by pass B; windows of 64 orphans (lane = orphan, ascending row order,
// position-major first fit as in p4_window) go to the waves round-robin.  Window w + 1 may look
// at live-list positions [j, j + U) as soon as window w is done with them (cell (orphan, node
// position) of the reference's double loop depends only on earlier orphans at that position and
// on earlier positions of that orphan), so consecutive windows run one step apart.
// prog[wave] = window << 32 | positions done (monotone; a finished window counts as the start of
// the next one).  A window waits for EVERY earlier window that may still be running (the latest
// window of each other wave; earlier windows of its own wave are finished), not only for its
// predecessor: that one can finish early while an older window still walks the list.  The earliest window that cannot place an orphan decides the failure
// (KAS:183-184): everything before it completed exactly as in the sequential order; later
// windows stop when they see it.  LDS words other waves write are read through a ballot or a
// broadcast, so a wave always acts on one answer.
// ---------------------------------------------------------------------------------------------
// node positions per hand-over step of the parallel P4.  The step is the chain (window w + 1 takes a
// position group when window w has published it) and its cost grows with the group: lists 5 wide test
// five holder racks per position and at configs[4] nearly every orphan lands on the first or second
// node of the group, so 2 positions (fill 3.4 ms) beat 4 (4.3) and 8 (5.2); at the headline shape
// (lists 3 wide, rack-conflict stragglers walking a list of few nodes) 4 is best (365k against 360k
// scenarios/s at 2 or 8).
#ifndef KAS_P4_U
#define KAS_P4_U 4
#endif
#ifndef KAS_P4_U_WIDE
#define KAS_P4_U_WIDE 1
#endif
// (NC: chunk lists the orphans come in — the fill's wavefronts — where that is not the number of wavefronts running the
// windows: kas_p4_kernel)
// (fin != nullptr — first fit in the order kernel's workgroup, kas_p4_order_kernel, NW == 1: behind every finished window the word
//  gets fin_hi | rows of the topic that are FINAL — every row below the next window's first orphan; the order wavefront of the
//  same workgroup follows it)
template <int W, int NW, int NC = NW>
KAS_DEV void p4_lists_parallel(const LdsView& L, const TopicView& T, int32_t live_count, int32_t wave,
                               int64_t (&st)[8], int32_t& fail_win, int32_t& fail_row, uint64_t* fin = nullptr, uint64_t fin_hi = 0ull) {
  const int lane = kasw::lane();
  uint64_t* prog = (uint64_t*)&L.ctl[KAS_CTL_PROG];
  int32_t oc[NC], total = 0;
#pragma unroll
  for (int w = 0; w < NC; ++w) { oc[w] = L.ctl[KAS_CTL_OC + w]; total += oc[w]; }
  // row index of the g-th orphan of the topic (chunk lists concatenated), or -1 past the end
  // row index of the g-th orphan of the topic (chunk lists concatenated), or -1 past the end
  auto orphan_row = [&](int32_t g) -> int32_t {
    int32_t w = 0, base = 0;
#pragma unroll
    for (int k = 0; k < NC - 1; ++k) {
      const bool next = w == k && g >= base + oc[k];
      base += next ? oc[k] : 0;
      w += next ? 1 : 0;
    }
    return g < total ? T.orph[((int64_t)chunk_begin<NC>(T.nt, w) << 6) + (g - base)] : -1;
  };
  auto row_cells = [&](int32_t p) -> MidRaw<W> {
    return mid_load_raw<W>(T.mid, T.ow, p >= 0 ? p : 0, p >= 0, T.m32);
  };
  const int32_t n_win = (total + 63) >> 6;
  // Window w may touch live-list positions [j, j + U) once EVERY earlier window is done with them.
  // Waiting for window w - 1 alone is not enough: it may finish early (its orphans all placed on the
  // first nodes) while window w - 2 still walks the list with an orphan whose racks were taken, and
  // window w would then overtake that orphan and take a slot that is not its turn (round 2: one
  // scenario solve in ~70,000 of the bench mix ended with a broker one over its cap).  Windows
  // w - NW and earlier ran on this wave and are finished; lane d (1 <= d < NW) watches the wave that
  // has window w - d.
  const int32_t dw = (lane >= 1 && lane < NW) ? lane : 1;
  const int32_t xw = (wave + NW - dw) % NW;
  const int32_t cap = T.cap, mw = mid_width(T.ow);
  constexpr int U = W >= 4 ? KAS_P4_U_WIDE : KAS_P4_U;      // node positions fetched per LDS round trip
  int32_t p_nxt = orphan_row(64 * wave + lane);
  MidRaw<W> c_nxt = row_cells(p_nxt);
  for (int32_t w = wave; w < n_win; w += NW) {
    const int32_t p = p_nxt;
    int32_t c_cur[W];
    mid_unpack<W>(c_nxt, T.ow, c_cur, T.m32);
    p_nxt = orphan_row(64 * (w + NW) + lane);              // my next window's rows: read ahead
    c_nxt = row_cells(p_nxt);
    kasw::repoll();
    if (kasw::ballot(L.ctl[KAS_CTL_FAILWIN] < w) != 0) break;   // an earlier window failed: so has the topic
    int32_t hc = 0, hr[W];                                  // holders are a prefix of the row
#pragma unroll
    for (int k = 0; k < W; ++k) {
      hr[k] = (p >= 0 && c_cur[k] >= 0) ? (int32_t)lds_rack(L, c_cur[k]) : -1;
      hc += (p >= 0 && c_cur[k] >= 0) ? 1 : 0;
    }
    int32_t need = p >= 0 ? T.rf - hc : 0;
    int32_t j = kasw::shfl(L.ctl[KAS_CTL_HEAD], 0);
    if (lane == 0) prog[wave] = ((uint64_t)(uint32_t)w << 32) | (uint32_t)j;
    KAS_COUNT(st[4]);
    bool stop = false;
    bool placed = false;                                    // (dword mid rows) my row took a broker in this window
    for (;;) {
      uint64_t pend = kasw::ballot(need > 0);
      if (pend == 0) break;
      if (j >= live_count) {                                // KAS:183-184: this orphan cannot be placed
        if (lane == 0) kasw::lds_atomic_min(&L.ctl[KAS_CTL_FAILWIN], w);
        stop = true;
        break;
      }
      // (the nodes of the position group and their racks do not change: read before the wait, so that what
      // follows it — the chain from window to window — is one LDS round trip for the loads)
      int32_t n[U], slots[U], rk[U];
#pragma unroll
      for (int u = 0; u < U; ++u) n[u] = (int32_t)L.live[j + u < live_count ? j + u : j];
#pragma unroll
      for (int u = 0; u < U; ++u) rk[u] = (int32_t)lds_rack(L, n[u]);
      if (w > 0 && NW > 1) {                                // until every earlier window is done with [j, j + U)
        const int32_t upto = j + U < live_count ? j + U : live_count;
        const bool watch = lane >= 1 && lane < NW && w - dw >= 0;
        const uint64_t want = ((uint64_t)(uint32_t)(w - dw) << 32) + (uint32_t)upto;
        bool abandoned = false;
        int32_t idle = 0;
        for (;;) {
          kasw::repoll();
          if (kasw::ballot(watch && prog[xw] < want) == 0) break;
          if (kasw::ballot(L.ctl[KAS_CTL_FAILWIN] < w) != 0) { abandoned = true; break; }
          if (watchdog_poll((uint32_t*)&L.ctl[KAS_CTL_WATCHDOG], false, idle)) {
            if (lane == 0) kasw::lds_atomic_min(&L.ctl[KAS_CTL_FAILWIN], -1);   // every later window stops
            abandoned = true;
            break;
          }
          // (no s_sleep between polls: the hand-over from window to window is the chain of P4, and the
          // poll is one LDS read; in flight 362.4k against 358.2k scenarios/s with the pause in round 3, 634k against
          // 653k in round 5)
        }
        if (abandoned) { stop = true; break; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) slots[u] = cap - lds_load(L, n[u]);
      int32_t taken[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        taken[u] = 0;
        if (j + u < live_count && pend != 0) {             // wave-uniform
          KAS_COUNT(st[5]);
          if (slots[u] > 0) {
            bool want = need > 0;
#pragma unroll
            for (int k = 0; k < W; ++k) want = want && !(k < hc && hr[k] == rk[u]);
            const uint64_t wm = kasw::ballot(want);
            if (wm != 0) {
              const int32_t rank = kasw::count_below(wm);
              if (want && rank < slots[u]) {               // accept (KAS:178-181)
                if (W == 3 && T.m32) {                       // (dword mid rows: the row is stored again, sorted, when its window is through)
                  put<W>(c_cur, hc, n[u]);
                  placed = true;
                } else {
                  T.mid[(int64_t)p * mw + hc] = (uint16_t)n[u];
                }
                put<W>(hr, hc, rk[u]);
                hc += 1;
                need -= 1;
              }
              const int32_t takers = kasw::popc(wm);
              taken[u] = takers < slots[u] ? takers : slots[u];
              pend = kasw::ballot(need > 0);
            }
          }
        }
      }
      if (lane == 0) {
#pragma unroll
        // (only this wave touches these nodes now: the new load follows from the slots read above, no re-read)
        for (int u = 0; u < U; ++u) if (taken[u] > 0) lds_load(L, n[u]) = cap - slots[u] + taken[u];
      }
      kasw::lockstep();
      j += U;
      if (lane == 0) prog[wave] = ((uint64_t)(uint32_t)w << 32) | (uint32_t)(j < live_count ? j : live_count);
    }
    if constexpr (W == 3) {
      if (T.m32 && placed) reinterpret_cast<uint32_t*>(T.mid)[p] = mid32_pack(c_cur[0], c_cur[1], c_cur[2]);
    }
    if (stop) {
      // failed or abandoned: whoever waits on this window must not hang
      if (j >= live_count) {
        const uint64_t left = kasw::ballot(need > 0);
        fail_win = w;
        fail_row = kasw::shfl(p, left != 0 ? kasw::first_lane(left) : 0);
      }
      if (lane == 0) prog[wave] = (uint64_t)(uint32_t)(w + 1) << 32;
      break;
    }
    // done: full nodes at the front of the live list need not be looked at again
    if (lane == 0) {
      int32_t head = L.ctl[KAS_CTL_HEAD];
      while (head < live_count && lds_load(L, (int32_t)L.live[head]) >= cap) ++head;
      kasw::lds_atomic_max(&L.ctl[KAS_CTL_HEAD], head);
      prog[wave] = (uint64_t)(uint32_t)(w + 1) << 32;
    }
    if (fin != nullptr) {                                    // (wave-uniform) this window's mid-row cells are out: publish
      const int32_t nxt0 = kasw::shfl(p_nxt, 0);             // the next window's first orphan (ascending rows), or none
      kasw::wave_sync();                                     // (release: the stores above before the word)
      if (lane == 0) kasw::store_shared_u64_lds(fin, fin_hi | (uint64_t)(uint32_t)(nxt0 >= 0 ? nxt0 : T.P));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// P5 helpers
// ---------------------------------------------------------------------------------------------
// The picks of one row (KAS:225-236).  Position k of the row's ascending node list has counter
// row c[k].  getLeastSeenNodeForReplicaId (KAS:263-278): the element of sorted rank i is visited
// at position (i + idx_m) % m; the first visited strictly smallest count wins, i.e. the minimum
// of (count, visit position).  pos[r] = list position picked for replica index r; cnt_r[r] = its
// counter value before the pick.
template <int W>
KAS_DEV void pick_row(const int32_t (&c)[W][W], int32_t Lp, bool valid, const int32_t (&idxm)[W + 1],
                      int32_t (&pos)[W], int32_t (&cnt_r)[W]) {
  uint32_t alive = valid ? ((1u << Lp) - 1u) : 0u;       // sorted-set positions still in nodeSet
#pragma unroll
  for (int r = 0; r < W; ++r) {
    const int32_t m = __builtin_popcount(alive);
    const int32_t idx = sel<W + 1>(idxm, m);
    // 64-bit keys: the counters are Java ints handed in with the Context — any value (count << 3 would wrap from 2^28)
    int64_t keys[W];
    int64_t best = 0x7fffffffffffffffll;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      int32_t rr = __builtin_popcount(alive & ((1u << k) - 1u)) + idx;
      rr -= rr >= m ? m : 0;
      // "alive bit k ? (count << 3 | rr) : INT64_MAX"
      keys[k] = ((alive >> k) & 1u) ? (((int64_t)c[k][r] << 3) | (int64_t)rr) : 0x7fffffffffffffffll;
      best = keys[k] < best ? keys[k] : best;
    }
    int32_t ps = 0;
#pragma unroll
    for (int k = 1; k < W; ++k) ps = keys[k] == best ? k : ps;
    pos[r] = ps;
    cnt_r[r] = (int32_t)(best >> 3);
    alive &= ~(1u << ps);                                // nodeSet.remove (KAS:232)
  }
}

// Sets.newTreeSet(preferenceList) (KAS:228): ascending node index == ascending broker id.
// cells: node indices or -1; result h[0..Lp) ascending.
template <int W>
KAS_DEV void sort_holders(const int32_t (&cells)[W], int32_t (&h)[W], int32_t& Lp) {
  Lp = 0;
#pragma unroll
  for (int k = 0; k < W; ++k) {
    h[k] = cells[k] >= 0 ? cells[k] : 0x7fffffff;
    Lp += cells[k] >= 0 ? 1 : 0;
  }
#pragma unroll
  for (int pass = 0; pass < W; ++pass) {
#pragma unroll
    for (int k = (pass & 1); k + 1 < W; k += 2) {
      const int32_t lo = h[k] < h[k + 1] ? h[k] : h[k + 1];
      const int32_t hi = h[k] < h[k + 1] ? h[k + 1] : h[k];
      h[k] = lo; h[k + 1] = hi;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fill kernel: one topic == P0-P4 of one getRackAwareAssignment call (+ tickets).  Executed by
// the whole workgroup.  Leaves the out rows holding node indices (holders in acceptance order,
// -1 padded); the order kernel turns them into the final preference lists.
// ---------------------------------------------------------------------------------------------
// SLIM (kas_fill_slim_kernel, round 6): the ONE path every BASELINE config at RF <= 3 takes — int32 cells, per-chunk histograms, a
// direct id table, the quota drawn with the atomic-with-return, first fit handed over — compiled without the other ~40 (general
// fill, binary search, chunk-count pass, 16-bit cells, index rows, first fit's windows): 120 VGPRs and NO scratch where the full
// kernel has 128 and 368 B per lane.  A topic that needs another path returns KAS_TOPIC_NEEDS_FULL_FILL: the scenario is handed
// back (KasLaunch::sp_flag) and the full kernel, launched behind this one for flagged scenarios only, solves it from its first topic.
#define KAS_TOPIC_NEEDS_FULL_FILL (-1000)
// (M32C: the mid-row layout as a compile-time constant — 1: dword mid rows (KAS_FLAG_MID32), 0: 16-bit rows, -1: what the launch's flags
//  say; the slim kernel is instantiated for each layout so that it holds one store path)
template <int W, int NW, bool SLIM = false, int M32C = -1>
KAS_DEV TopicOutcome fill_topic(const KasLaunch& a, const kas_topic_desc& td, const LdsView& L,
                                const NodeMap& nm, const int32_t* g_node_id, const int32_t* g_node_rack,
                                uint64_t* accmask, int32_t* orph, int32_t* p4s, int64_t (&st)[8]) {
  constexpr int NT = 64 * NW;
  const int lane = kasw::lane();
  const int tid = kasw::tid();
  const int32_t wave = kasw::wave_id();
  const uint64_t lt = kasw::lanemask_lt();
  const int32_t N = nm.n;
  TopicView T;
  T.c16 = cells16(a);
  T.cur = topic_cur(a, td);
  T.orph = orph;
  T.len_arr = td.cur_len_off >= 0 ? a.aux + td.cur_len_off : nullptr;
  T.inp_arr = td.in_partitions_off >= 0 ? a.aux + td.in_partitions_off : nullptr;
  T.pid_arr = td.part_id_off >= 0 ? a.aux + td.part_id_off : nullptr;
  T.P = td.n_partitions; T.cw = td.cur_width; T.rf = td.rf; T.ow = td.out_width;
  T.hash = td.name_hash; T.nt = (T.P + 63) >> 6; T.N = N;
  T.mid = topic_mid(a, td);
  T.m32 = M32C < 0 ? mid32(a) : (M32C != 0);
  const int32_t P = T.P, hash = T.hash;

  TopicOutcome res;
  res.status = KAS_OK; res.fail_partition = -1;
  res.moved_replicas = 0; res.moved_partitions = 0;

  int64_t tmark = kasw::clock_ticks();
  // ---- P0: capacity (KAS:45, 65-71) over |partitions| ---------------------------------------
  int32_t n_in = P;
  if (T.inp_arr) {                                          // every wave counts for itself
    int32_t c = 0;
    for (int32_t p = lane; p < P; p += 64) c += T.inp_arr[p] != 0 ? 1 : 0;
    n_in = kasw::wave_sum(c);
  }
  T.cap = max_replicas_per_node(N, n_in, T.rf);
  const int32_t cap = T.cap;

  // ---- per-topic node state (KAS:46, 73-99): region A of the LDS carve-up -------------------
  // rack-diverse form of the sticky fill unless switched off or the quota word cannot hold cap
  // (a topic without rows takes the general form: the row stream of the fast form re-reads the last
  // row for lanes past the end and there is no row to read)
  const bool try_fast = P > 0 && T.cw > 0 && !(a.flags & KAS_FLAG_GENERIC_FILL) && cap >= 0 && cap < (1 << 28);
  // per-chunk histograms (no chunk-count pass) when the launcher says the layout fits (lists <= 3 wide)
  const bool fused = W <= 3 && NW > 1 && (a.flags & KAS_FLAG_FUSED_HIST) != 0u;
  if (fused) {
    // the node's block: counts (and, in the same words later, quotas, load, qrs) zero, then its rack
    constexpr int BW = fused_block_words<(W <= 3 ? W : 3), (NW > 1 ? NW : 2)>();
    for (int32_t i = tid; i < N; i += NT) {
#pragma unroll
      for (int k = 0; k < BW - 1; ++k) L.x[i * BW + k] = 0;
      lds_rack(L, i) = (int16_t)g_node_rack[i];
    }
  } else {
    for (int32_t i = tid; i < N; i += NT) {
      lds_load(L, i) = 0;
      lds_rack(L, i) = (int16_t)g_node_rack[i];
    }
    if (try_fast)
      for (int32_t i = tid; i < N * W; i += NT) L.x[i] = 0;
  }
  if (nm.range != 0u) {
    for (uint32_t i = (uint32_t)tid; i < nm.range; i += (uint32_t)NT) L.idmap[i] = (int16_t)-1;
  } else {
    for (int32_t i = tid; i < N; i += NT) L.ids[i] = g_node_id[i];
  }
  if (tid < KAS_CTL_INTS) L.ctl[tid] = tid == KAS_CTL_FAILROW ? -1 : (tid == KAS_CTL_FAILWIN ? 0x7fffffff : 0);
  kasw::sync();
  if (nm.range != 0u)
    for (int32_t i = tid; i < N; i += NT) L.idmap[(uint32_t)g_node_id[i] - (uint32_t)nm.min_id] = (int16_t)i;
  kasw::sync();
  { const int64_t now = kasw::clock_ticks(); st[0] += now - tmark; tmark = now; }

  // ---- P2: sticky fill (KAS:49, 101-131) ----------------------------------------------------
  if constexpr (SLIM) {
    static_assert(W <= 3 && NW > 1, "the slim fill is the per-chunk-histogram form");
    const bool mine = try_fast && fused && !T.c16 && (a.flags & KAS_FLAG_LANE_ORDER) != 0u && !(a.flags & KAS_FLAG_INDEX_ROWS) &&
                      nm.range != 0u && p4s != nullptr;
    if (!mine) { res.status = KAS_TOPIC_NEEDS_FULL_FILL; return res; }      // (workgroup-uniform)
    const bool viol = fill_pass_a_fused<W, NW, true>(L, T, nm, wave);
    if (kasw::ballot(viol) != 0 && lane == 0) L.ctl[KAS_CTL_VIOL] = 1;
    kasw::sync();
    if (L.ctl[KAS_CTL_VIOL] != 0) { res.status = KAS_TOPIC_NEEDS_FULL_FILL; return res; }   // rows not rack-diverse: the general fill's case
    fill_quota_fused<W, NW>(L, T, tid);
    kasw::sync();
    { const int64_t now = kasw::clock_ticks(); st[1] += now - tmark; tmark = now; }
    for (int32_t i = tid; i < N; i += NT)                     // (pass B's one 16-bit word per cell: node index | r* << 14, as below)
      L.idmap[(uint32_t)g_node_id[i] - (uint32_t)nm.min_id] =
          (int16_t)(uint16_t)((uint32_t)i | (((uint32_t)lds_qrs(L, i) >> 28) << 14));
    kasw::sync();
    int32_t moved_r = 0, moved_p = 0;
    const int32_t oc = fill_pass_b<W, NW, true, true, true>(L, T, nm, wave, moved_r, moved_p, st);
    if (lane == 0) L.ctl[KAS_CTL_OC + wave] = oc;
    kasw::sync();
    { const int64_t now = kasw::clock_ticks(); st[2] += now - tmark; tmark = now; }
    if (java_abs_mod(hash, N) < 0) { res.status = KAS_FAIL_HASH_INDEX; return res; }   // KAS:168 (workgroup-uniform)
    for (int32_t i = tid; i < N; i += NT) p4s[KAS_P4S_HEAD + i] = lds_load(L, i);     // the hand-over to first fit, as below
    if (tid == 0) { p4s[0] = 1; p4s[1] = cap; }
    if (tid < NW) p4s[2 + tid] = L.ctl[KAS_CTL_OC + tid];
    const int32_t mr = kasw::wave_sum(moved_r), mp = kasw::wave_sum(moved_p);
    if (lane == 0 && (mr | mp) != 0) {
      kasw::lds_atomic_add(&L.ctl[KAS_CTL_MOVED_R], mr);
      kasw::lds_atomic_add(&L.ctl[KAS_CTL_MOVED_P], mp);
    }
    kasw::sync();
    { const int64_t now = kasw::clock_ticks(); st[3] += now - tmark; tmark = now; }
    res.moved_replicas = L.ctl[KAS_CTL_MOVED_R];
    res.moved_partitions = L.ctl[KAS_CTL_MOVED_P];
    return res;
  } else {
  bool fast = false;
  // index rows (KAS_FLAG_INDEX_ROWS): int32 cells, per-chunk histograms, the quota drawn with the atomic-with-return, a direct
  // id table and rows exactly W wide — pass A leaves the rows' node indices where the mid rows go and pass B streams those
  // (workgroup-uniform; 14-bit node indices in the rewritten table below)
  const bool ixrows = W <= 3 && NW > 1 && try_fast && fused && !T.c16 && (a.flags & KAS_FLAG_INDEX_ROWS) != 0u &&
                      (a.flags & KAS_FLAG_LANE_ORDER) != 0u && nm.range != 0u && full_rows_of<W>(T) && N < 0x3fff;
  if (try_fast) {
    bool viol;
    if constexpr (W <= 3 && NW > 1) {
      if (fused) viol = ixrows ? fill_pass_a_fused<W, NW, true, false, true>(L, T, nm, wave)
                        : (T.c16 && nm.range != 0u && nm.min_id == 0) ? fill_pass_a_fused<W, NW, true, true>(L, T, nm, wave)
                        : (nm.range != 0u ? fill_pass_a_fused<W, NW, true>(L, T, nm, wave) : fill_pass_a_fused<W, NW, false>(L, T, nm, wave));
      else viol = nm.range != 0u ? fill_pass_a<W, NW, true>(L, T, nm, wave) : fill_pass_a<W, NW, false>(L, T, nm, wave);
    } else {
      viol = nm.range != 0u ? fill_pass_a<W, NW, true>(L, T, nm, wave) : fill_pass_a<W, NW, false>(L, T, nm, wave);
    }
    if (kasw::ballot(viol) != 0 && lane == 0) L.ctl[KAS_CTL_VIOL] = 1;
    kasw::sync();
    fast = L.ctl[KAS_CTL_VIOL] == 0;
    if (!fast) {
      // the general fill starts from load == 0; for wide lists load[] shares LDS with a histogram row
      for (int32_t i = tid; i < N; i += NT) lds_load(L, i) = 0;
      kasw::sync();
    }
  }
  int32_t moved_r = 0, moved_p = 0;
  if (fast) {
    int32_t oc = 0;
    bool done = false;
    if constexpr (W <= 3 && NW > 1) {
      if (fused) {                                          // workgroup-uniform
        fill_quota_fused<W, NW>(L, T, tid);
        kasw::sync();
        { const int64_t now = kasw::clock_ticks(); st[1] += now - tmark; tmark = now; }
        if (ixrows) {                                          // (workgroup-uniform)
          st[6] += 1;                                            // (kas_plan_stats()[6] of a fill launch: topics that took index rows)
          // pass B over the index rows pass A left in the mid region, as a topic of 16-bit cells whose node i has id i: the
          // table it looks into is node index -> node index | r* << 14 (the broker ids are not needed again for this topic)
          for (int32_t i = tid; i < N; i += NT)
            L.idmap[i] = (int16_t)(uint16_t)((uint32_t)i | (((uint32_t)lds_qrs(L, i) >> 28) << 14));
          kasw::sync();
          TopicView T2 = T;
          T2.c16 = true;
          T2.cur = reinterpret_cast<const int32_t*>(T.mid);
          NodeMap nm2;
          nm2.n = N; nm2.min_id = 0; nm2.range = (uint32_t)N;
          oc = fill_pass_b<W, NW, true, true, true, true>(L, T2, nm2, wave, moved_r, moved_p, st);
        }
        else if (a.flags & KAS_FLAG_LANE_ORDER) {              // (workgroup-uniform)
          if (nm.range != 0u) {
            // pass B reads ONE 16-bit word per cell: node index | r* << 14 (no node has index 0x3fff: the LDS ends
            // at 13,492 brokers; an id that is no broker keeps 0xffff) — the id table is not looked up again for
            // this topic (P4 works on node indices)
            for (int32_t i = tid; i < N; i += NT)
              L.idmap[(uint32_t)g_node_id[i] - (uint32_t)nm.min_id] =
                  (int16_t)(uint16_t)((uint32_t)i | (((uint32_t)lds_qrs(L, i) >> 28) << 14));
            kasw::sync();
            oc = fill_pass_b<W, NW, true, true, true>(L, T, nm, wave, moved_r, moved_p, st);
          } else {
            oc = fill_pass_b<W, NW, false, true, true>(L, T, nm, wave, moved_r, moved_p, st);
          }
        }
        else
          oc = nm.range != 0u ? fill_pass_b<W, NW, true, true>(L, T, nm, wave, moved_r, moved_p, st)
                              : fill_pass_b<W, NW, false, true>(L, T, nm, wave, moved_r, moved_p, st);
        done = true;
      }
    }
    if (!done) {
      fill_quota<W, NW>(L, T, tid);
      kasw::sync();
      if (NW > 1) {
        if (nm.range != 0u) fill_chunk_count<W, NW, true>(L, T, nm, wave);
        else fill_chunk_count<W, NW, false>(L, T, nm, wave);
        kasw::sync();
        fill_chunk_prefix<NW>(L, T, tid);
        kasw::sync();
      }
      { const int64_t now = kasw::clock_ticks(); st[1] += now - tmark; tmark = now; }
      oc = nm.range != 0u ? fill_pass_b<W, NW, true>(L, T, nm, wave, moved_r, moved_p, st)
                          : fill_pass_b<W, NW, false>(L, T, nm, wave, moved_r, moved_p, st);
    }
    if (lane == 0) L.ctl[KAS_CTL_OC + wave] = oc;
  } else if (wave == 0) {
    fill_generic_sweeps<W>(L, T, nm, accmask, st);
    { const int64_t now = kasw::clock_ticks(); st[1] += now - tmark; tmark = now; }
  }
  kasw::sync();
  { const int64_t now = kasw::clock_ticks(); st[2] += now - tmark; tmark = now; }

  // ---- KAS:168: getNodeProcessingOrder(topic, all nodes); runs even with zero orphans -------
  const int32_t idxN = java_abs_mod(hash, N);
  if (idxN < 0) { res.status = KAS_FAIL_HASH_INDEX; return res; }      // workgroup-uniform
  // ---- split first fit (KAS_FLAG_SPLIT_P4): a rack-diverse topic ends here — the brokers' loads, the orphan counts of the
  // chunks and cap go to kas_p4_kernel, which runs P4 (KAS:56, 162-186) on a quarter of this workgroup's LDS and registers
  // and reports a partition that cannot be placed; the topic's result says OK until then
  if (p4s != nullptr) {                                      // (workgroup-uniform)
    if (fast) {
      for (int32_t i = tid; i < N; i += NT) p4s[KAS_P4S_HEAD + i] = lds_load(L, i);
      if (tid == 0) { p4s[0] = 1; p4s[1] = cap; }
      if (tid < NW) p4s[2 + tid] = L.ctl[KAS_CTL_OC + tid];
      const int32_t mr = kasw::wave_sum(moved_r), mp = kasw::wave_sum(moved_p);
      if (lane == 0 && (mr | mp) != 0) {
        kasw::lds_atomic_add(&L.ctl[KAS_CTL_MOVED_R], mr);
        kasw::lds_atomic_add(&L.ctl[KAS_CTL_MOVED_P], mp);
      }
      kasw::sync();
      { const int64_t now = kasw::clock_ticks(); st[3] += now - tmark; tmark = now; }
      res.moved_replicas = L.ctl[KAS_CTL_MOVED_R];
      res.moved_partitions = L.ctl[KAS_CTL_MOVED_P];
      return res;
    }
    if (tid == 0) p4s[0] = 0;                                // (the general fill: first fit below, in this workgroup)
  }
  if (wave == 0) {
    const int32_t start = (N - idxN) % N;        // order[j] = sorted[(j + start) % N]
    // non-full nodes in processing order (full nodes can never accept again)
    int32_t live_count = 0;
    for (int32_t base = 0; base < N; base += 64) {
      const int32_t j = base + lane;
      int32_t n = j + start; if (n >= N) n -= N;
      const bool is_live = j < N && lds_load(L, n) < cap;
      const uint64_t m = kasw::ballot(is_live);
      if (is_live) L.live[live_count + kasw::count_below(m)] = (int16_t)n;
      live_count += kasw::popc(m);
    }
    kasw::lockstep();
    if (lane == 0) L.ctl[KAS_CTL_LIVE] = live_count;
    // ---- P3 + P4: orphans (KAS:52, 133-160) and first fit (KAS:56, 162-186) -----------------
    if (!fast) {
      const int32_t fail_row = p3p4_generic<W>(L, T, nm, accmask, live_count, moved_r, moved_p, st);
      if (lane == 0) L.ctl[KAS_CTL_FAILROW] = fail_row;
    }
  }
  if (fast) {                                                // workgroup-uniform: every wave takes windows
    kasw::sync();
    int32_t fail_win = -1, fail_row = -1;
    p4_lists_parallel<W, NW>(L, T, L.ctl[KAS_CTL_LIVE], wave, st, fail_win, fail_row);
    kasw::sync();                                            // KAS_CTL_FAILWIN is final: its wave reports the row
    if (fail_win >= 0 && fail_win == L.ctl[KAS_CTL_FAILWIN] && lane == 0) L.ctl[KAS_CTL_FAILROW] = fail_row;
    if (KAS_SPIN_BOUND > 0 && L.ctl[KAS_CTL_WATCHDOG] != 0) { res.status = KAS_FAIL_WATCHDOG; return res; }   // workgroup-uniform
  }
  {
    const int32_t mr = kasw::wave_sum(moved_r), mp = kasw::wave_sum(moved_p);
    if (lane == 0 && (mr | mp) != 0) {
      kasw::lds_atomic_add(&L.ctl[KAS_CTL_MOVED_R], mr);
      kasw::lds_atomic_add(&L.ctl[KAS_CTL_MOVED_P], mp);
    }
  }
  kasw::sync();   // out rows of P3/P4 are visible to every wave; load/qrs are dead from here
  { const int64_t now = kasw::clock_ticks(); st[3] += now - tmark; tmark = now; }
  {
    const int32_t fail_row = L.ctl[KAS_CTL_FAILROW];
    if (fail_row >= 0) {                                       // KAS:183-184
      res.status = KAS_FAIL_UNASSIGNABLE;
      res.fail_partition = T.pid_arr ? T.pid_arr[fail_row] : fail_row;
      return res;
    }
  }
  res.moved_replicas = L.ctl[KAS_CTL_MOVED_R];
  res.moved_partitions = L.ctl[KAS_CTL_MOVED_P];

  return res;
  }                                                          // (!SLIM)
}

// ---------------------------------------------------------------------------------------------
// fill kernel, one scenario: the per-topic loop of KAG:173-184 up to (not including) P5.
// Writes the topic results and the scenario record (digest 0; the order kernel completes it).
// ---------------------------------------------------------------------------------------------
template <int W, int NW, bool SLIM = false, int M32C = -1>
KAS_DEV void fill_scenario(const KasLaunch& a, int32_t s, unsigned char* lds_raw) {
  constexpr int NT = 64 * NW;
  const int tid = kasw::tid();
  if (!SLIM && (a.flags & KAS_FLAG_ONLY_FLAGGED) && a.sp_flag[s] == 0) return;   // the spread fill / the slim fill kernel did this one
  if (SLIM && tid == 0) a.sp_flag[s] = 0;                    // (this thread alone writes the flag: 1 below if the scenario is handed back)
  const kas_scenario_desc sd = a.scen[s];
  const int32_t N = sd.n_nodes;
  const KasLds lay = SLIM ? kas_fill_slim_lds(a.n_max, W, a.idmap_entries)
                          : kas_fill_lds_layout(a.n_max, W, NW, a.idmap_entries, a.need_bsearch,
                                                (a.flags & KAS_FLAG_GENERIC_FILL) ? 0 : ((a.flags & KAS_FLAG_FUSED_HIST) ? 2 : 1));
  LdsView L;
  L.x = (int32_t*)(lds_raw + lay.off_x);
  // (wide lists: load[] takes the place of histogram row NW of THIS scenario's node count once the
  // quota pass has consumed it, see kas_fill_lds_layout)
  L.load = (W > NW && !(a.flags & (KAS_FLAG_GENERIC_FILL | KAS_FLAG_FUSED_HIST))) ? L.x + NW * N : (int32_t*)(lds_raw + lay.off_load);
  L.qrs = (int32_t*)(lds_raw + lay.off_qrs);
  L.rack = (int16_t*)(lds_raw + lay.off_rack);
  L.live = (int16_t*)(lds_raw + lay.off_live);
  // fused layout: load, qrs and rack are words of the node's own block (kas_fill_lds_layout)
  L.ns = (W <= 3 && NW > 1 && (a.flags & KAS_FLAG_FUSED_HIST) && !(a.flags & KAS_FLAG_GENERIC_FILL)) ? kas_fused_block_words(W, NW) : 1;
  L.rs = L.ns > 1 ? 2 * L.ns : 1;
  L.idmap = (int16_t*)(lds_raw + lay.off_idmap);
  L.ids = (int32_t*)(lds_raw + lay.off_ids);
  L.ring_p = (int32_t*)(lds_raw + lay.off_ring);
  L.ring_meta = L.ring_p + KAS_RING_CAP;
  L.ring_rack = (int16_t*)(L.ring_meta + KAS_RING_CAP);
  L.ctl = (int32_t*)(lds_raw + lay.off_ctl);

  const int32_t* g_node_id = a.node_id + sd.node_off;
  const int32_t* g_node_rack = a.node_rack + sd.node_off;

  int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t t_begin = kasw::clock_ticks();
  // node table checks: strictly ascending, non-negative ids; racks in int16 range
  bool bad = false;
  for (int32_t i = tid; i < N; i += NT) {
    const int32_t id = g_node_id[i];
    const int32_t prev = i > 0 ? g_node_id[i - 1] : -1;
    const int32_t rk = g_node_rack[i];
    bad = bad || id <= prev || rk < 0 || rk > 32767;
  }
  if (tid == 0) L.ctl[KAS_CTL_VIOL] = 0;
  kasw::sync();
  if (kasw::ballot(bad) != 0 && kasw::lane() == 0) L.ctl[KAS_CTL_VIOL] = 1;
  kasw::sync();
  const bool nodes_bad = L.ctl[KAS_CTL_VIOL] != 0;
  NodeMap nm;
  nm.n = N; nm.min_id = 0; nm.range = 0u;
  if (N > 0 && !nodes_bad) {
    const int64_t lo = g_node_id[0], hi = g_node_id[N - 1];
    const int64_t range = hi - lo + 1;
    nm.min_id = (int32_t)lo;
    if (range <= (int64_t)a.idmap_entries) nm.range = (uint32_t)range;
  }
  kasw::sync();

  st[0] += kasw::clock_ticks() - t_begin;
  uint64_t* accmask = a.accmask + a.accmask_off[s];
  int32_t* orph = a.orph + a.orph_off[s];                   // (the next topic's stretch)
  int32_t scen_status = KAS_OK, fail_topic = -1, fail_part = -1;
  int32_t moved_r = 0, moved_p = 0;
  for (int32_t k = 0; k < sd.topic_count; ++k) {
    const int32_t ti = sd.topic_begin + k;
    const kas_topic_desc td = a.topics[ti];
    int32_t* const orph_topic = orph;                                  // (every topic its own stretch of the orphan scratch)
    orph += (int64_t)((td.n_partitions > 0 ? td.n_partitions : 0) + 63) / 64 * 64;
    int32_t* const p4s = (a.flags & KAS_FLAG_SPLIT_P4) ? a.p4s + (int64_t)ti * (KAS_P4S_HEAD + a.n_max) : nullptr;
    if (p4s != nullptr && tid == 0) p4s[0] = 0;                        // (until a rack-diverse topic says otherwise)
    TopicOutcome o;
    o.status = KAS_OK; o.fail_partition = -1; o.moved_replicas = 0; o.moved_partitions = 0;
    if (scen_status != KAS_OK) o.status = KAS_SKIPPED;                 // KAG:173-184 aborted
    else if (nodes_bad) o.status = KAS_FAIL_BAD_NODES;
    else if (!(td.rf > 0)) o.status = KAS_FAIL_RF_NOT_POSITIVE;        // KTA:65-66
    else if (!(td.rf <= N)) o.status = KAS_FAIL_RF_GT_BROKERS;         // KTA:67-69
    else o = fill_topic<W, NW, SLIM, M32C>(a, td, L, nm, g_node_id, g_node_rack, accmask, orph_topic, p4s, st);
    if constexpr (SLIM) {
      if (o.status == KAS_TOPIC_NEEDS_FULL_FILL) {            // (workgroup-uniform) hand the scenario back: the full kernel solves it from its first topic
        if (tid == 0) a.sp_flag[s] = 1;
        return;
      }
    }
    kasw::sync();
    if (o.status != KAS_OK) {
      // nothing is returned for a failed topic: its rows are all padding
      out_pad(topic_out(a, td), (int64_t)td.n_partitions * td.out_width, tid, NT);
      o.moved_replicas = 0; o.moved_partitions = 0;
      if (scen_status == KAS_OK) { scen_status = o.status; fail_topic = k; fail_part = o.fail_partition; }
    }
    if (tid == 0) {
      kas_topic_result tr;
      tr.status = o.status; tr.fail_partition = o.fail_partition;
      tr.moved_replicas = o.moved_replicas; tr.moved_partitions = o.moved_partitions;
      a.topic_results[ti] = tr;
    }
    moved_r += o.moved_replicas; moved_p += o.moved_partitions;
    kasw::sync();
  }
  if (tid == 0) {
    kas_scenario_result sr;
    sr.status = scen_status; sr.fail_topic = fail_topic; sr.fail_partition = fail_part;
    sr.moved_replicas = moved_r; sr.moved_partitions = moved_p; sr.reserved = 0;
    sr.digest = 0;
    a.scenario_results[s] = sr;
    if (a.stats) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a.stats[(int64_t)s * KAS_STATS_PER_SCENARIO + i] = st[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// kas_fill_kernel, one workgroup: the scenarios workgroup `block` of `grid` takes.  A launch for every scenario deals them by
// index (block, block + grid, ...).  A launch for FLAGGED scenarios only (KAS_FLAG_ONLY_FLAGGED with the flags in sp_flag: behind
// the slim kernel, behind the spread fill, the wide form's second solve) deals them BY RANK among the flagged ones — workgroup b
// takes the b-th, (b + grid)-th, ... flagged scenario — so that the scenarios handed back spread evenly over the workgroups
// however they lie among the others, and workgroup 0 leaves their number in KasLaunch::handback (when given: host memory the
// plan sizes its next such launch by).
// ---------------------------------------------------------------------------------------------
// The r-th scenario (ascending) whose flag is set, or -1 past the last one; total: their number when the answer is -1.  Every
// wavefront of the workgroup scans the flags for itself (64 per step, a ballot and a population count: wave-uniform by
// construction, nothing shared, no barrier) and gets the same answer: nothing writes the flags while this kernel runs.
KAS_DEV int32_t kth_flagged(const int32_t* flag, int32_t n, int32_t r, int32_t& total) {
  const int32_t lane = kasw::lane();
  int32_t seen = 0;
  for (int32_t base = 0; base < n; base += 64) {
    const int32_t s = base + lane;
    uint64_t w = kasw::ballot(s < n && flag[s < n ? s : 0] != 0);
    const int32_t c = kasw::popc(w);
    if (r < seen + c) {
      for (int32_t i = seen; i < r; ++i) w &= w - 1ull;      // (drop the r - seen lowest set bits)
      return base + kasw::first_lane(w);
    }
    seen += c;
  }
  total = seen;
  return -1;
}

template <int W, int NW>
KAS_DEV void fill_block(const KasLaunch& a, int32_t block, int32_t grid, unsigned char* lds_raw) {
  const bool by_rank = (a.flags & KAS_FLAG_ONLY_FLAGGED) != 0u && a.sp_flag != nullptr;
  for (int32_t i = block;; i += grid) {
    int32_t s = i, total = 0;
    if (by_rank) {
      s = kth_flagged(a.sp_flag, a.n_scenarios, i, total);
      if (s < 0) {
        if (block == 0 && kasw::tid() == 0 && a.handback != nullptr) *a.handback = total;
        break;
      }
    } else if (s >= a.n_scenarios) {
      break;
    }
    fill_scenario<W, NW>(a, s, lds_raw);
  }
}

// ---------------------------------------------------------------------------------------------
// kas_p4_kernel, one scenario (KAS_FLAG_SPLIT_P4): first fit (P4, KAS:56, 162-186) of the topics the fill kernel handed
// over, in order, on four wavefronts — p4_lists_parallel exactly as the fill workgroup ran it, on the loads the sticky
// fill left (KasLaunch::p4s), the topic's orphan lists and its mid rows.  A partition that cannot be placed fails its
// topic (KAS:183-184), the topics behind it are skipped (KAG:173-184 aborted) and emit nothing, and the scenario's
// record says so — what fill_scenario does when first fit runs inside it.
// ---------------------------------------------------------------------------------------------
// FS (kas_p4_order_kernel: first fit as ONE wavefront of the order kernel's workgroup, whatever its index there): the workgroup
// barriers become wavefront barriers, and fs[] carries what the order wavefront follows —
//   fs[0]  topic << 32 | rows of that topic that are final (first fit done with them; a topic that needs none: all its rows)
//   fs[1]  the topic first fit failed at (KAS:183-184), or 0x7fffffff;  fs[2]  the order wavefront's answer: it has stopped writing
// — and a failed topic's padding waits for that answer (the order wavefront may have emitted rows of it already).
// (M32C: the mid-row layout as a compile-time constant, as in fill_topic — 1: dword mid rows, 0: 16-bit rows, -1: the launch's flags)
template <int W, int PW, bool FS = false, int M32C = -1>
KAS_DEV void p4_scenario(const KasLaunch& a, int32_t s, unsigned char* lds_raw, uint64_t* fs = nullptr) {
  constexpr int NW = PW, NT = 64 * NW, NC = KAS_P4_WAVES;    // PW wavefronts run the windows over the fill's NC chunk lists
  static_assert(!FS || PW == 1, "first fit inside the order kernel's workgroup is one wavefront");
  const int lane = kasw::lane();
  const int tid = FS ? lane : kasw::tid();
  const int32_t wave = FS ? 0 : kasw::wave_id();
  auto barrier = [&]() { if constexpr (FS) kasw::wave_sync(); else kasw::sync(); };
  const kas_scenario_desc sd = a.scen[s];
  const int32_t N = sd.n_nodes;
  // (a scenario the fill kernel failed at topic k2 still has its rack-diverse topics before k2 waiting for their first fit)
  const KasP4Lds lay = kas_p4_lds_layout(a.n_max);
  LdsView L;
  L.x = nullptr; L.qrs = nullptr; L.idmap = nullptr; L.ids = nullptr; L.ring_p = nullptr; L.ring_meta = nullptr; L.ring_rack = nullptr;
  L.load = (int32_t*)(lds_raw + lay.off_load);
  L.rack = (int16_t*)(lds_raw + lay.off_rack);
  L.live = (int16_t*)(lds_raw + lay.off_live);
  L.ctl = (int32_t*)(lds_raw + lay.off_ctl);
  L.ns = 1; L.rs = 1;
  const int32_t* g_node_rack = a.node_rack + sd.node_off;
  const int64_t t_begin = kasw::clock_ticks();
  int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int32_t* orph = a.orph + a.orph_off[s];
  bool failed = false;
  int32_t fail_topic = -1, fail_part = -1, moved_r = 0, moved_p = 0;   // (moved: over the topics before a failure)
  for (int32_t k = 0; k < sd.topic_count; ++k) {
    const int32_t ti = sd.topic_begin + k;
    const kas_topic_desc td = a.topics[ti];
    int32_t* const orph_topic = orph;
    orph += (int64_t)((td.n_partitions > 0 ? td.n_partitions : 0) + 63) / 64 * 64;
    if (failed) {                                            // KAG:173-184 aborted: nothing is returned for this topic
      // (FS: the order wavefront skips every topic behind fs[1] and never writes there)
      out_pad(topic_out(a, td), (int64_t)td.n_partitions * td.out_width, tid, NT);
      if (tid == 0) {
        kas_topic_result tr;
        tr.status = KAS_SKIPPED; tr.fail_partition = -1; tr.moved_replicas = 0; tr.moved_partitions = 0;
        a.topic_results[ti] = tr;
      }
      continue;
    }
    const kas_topic_result tr0 = a.topic_results[ti];
    moved_r += tr0.moved_replicas; moved_p += tr0.moved_partitions;
    const int32_t* p4s = a.p4s + (int64_t)ti * (KAS_P4S_HEAD + a.n_max);
    if (tr0.status != KAS_OK || p4s[0] == 0) {               // (workgroup-uniform: nothing handed over)
      if constexpr (FS) {                                    // every row of the topic is final as the fill kernel left it
        if (lane == 0) kasw::store_shared_u64_lds(&fs[0], ((uint64_t)(uint32_t)k << 32) | (uint64_t)(uint32_t)(td.n_partitions > 0 ? td.n_partitions : 0));
      }
      continue;
    }
    TopicView T;
    T.c16 = cells16(a); T.cur = nullptr; T.orph = orph_topic; T.len_arr = nullptr; T.inp_arr = nullptr;
    T.pid_arr = td.part_id_off >= 0 ? a.aux + td.part_id_off : nullptr;
    T.P = td.n_partitions; T.cw = td.cur_width; T.rf = td.rf; T.ow = td.out_width;
    T.hash = td.name_hash; T.nt = (T.P + 63) >> 6; T.N = N;
    T.mid = topic_mid(a, td);
    T.m32 = M32C < 0 ? mid32(a) : (M32C != 0);
    T.cap = p4s[1];
    const int32_t cap = T.cap;
    barrier();                                            // (the previous topic's node state has been read)
    for (int32_t i = tid; i < N; i += NT) { L.load[i] = p4s[KAS_P4S_HEAD + i]; L.rack[i] = (int16_t)g_node_rack[i]; }
    if (tid < KAS_CTL_INTS) L.ctl[tid] = tid == KAS_CTL_FAILROW ? -1 : (tid == KAS_CTL_FAILWIN ? 0x7fffffff : 0);
    barrier();
    if (tid < NC) L.ctl[KAS_CTL_OC + tid] = p4s[2 + tid];
    if (wave == 0) {                                         // non-full nodes in processing order (KAS:168, 188-200), as fill_topic
      const int32_t idxN = java_abs_mod(T.hash, N);          // (>= 0: the fill kernel checked)
      const int32_t start = (N - idxN) % N;
      int32_t live_count = 0;
      for (int32_t base = 0; base < N; base += 64) {
        const int32_t j = base + lane;
        int32_t n = j + start; if (n >= N) n -= N;
        const bool is_live = j < N && lds_load(L, n) < cap;
        const uint64_t m = kasw::ballot(is_live);
        if (is_live) L.live[live_count + kasw::count_below(m)] = (int16_t)n;
        live_count += kasw::popc(m);
      }
      kasw::lockstep();
      if (lane == 0) L.ctl[KAS_CTL_LIVE] = live_count;
    }
    barrier();
    int32_t fail_win = -1, fail_row = -1;
    if constexpr (FS) {
      // rows below the topic's first orphan are final already
      const int32_t total_o = L.ctl[KAS_CTL_OC] + L.ctl[KAS_CTL_OC + 1] + L.ctl[KAS_CTL_OC + 2] + L.ctl[KAS_CTL_OC + 3];
      int32_t first = T.P;
      if (total_o > 0) {
        int32_t w0 = 0;
        while (w0 < NC - 1 && L.ctl[KAS_CTL_OC + w0] == 0) ++w0;
        first = T.orph[(int64_t)chunk_begin<NC>(T.nt, w0) << 6];
      }
      if (lane == 0) kasw::store_shared_u64_lds(&fs[0], ((uint64_t)(uint32_t)k << 32) | (uint64_t)(uint32_t)first);
      p4_lists_parallel<W, NW, NC>(L, T, L.ctl[KAS_CTL_LIVE], wave, st, fail_win, fail_row, &fs[0], (uint64_t)(uint32_t)k << 32);
    } else {
      p4_lists_parallel<W, NW, NC>(L, T, L.ctl[KAS_CTL_LIVE], wave, st, fail_win, fail_row);
    }
    barrier();                                            // KAS_CTL_FAILWIN is final: its wave reports the row
    if (fail_win >= 0 && fail_win == L.ctl[KAS_CTL_FAILWIN] && lane == 0) L.ctl[KAS_CTL_FAILROW] = fail_row;
    barrier();
    const bool hung = KAS_SPIN_BOUND > 0 && L.ctl[KAS_CTL_WATCHDOG] != 0;
    const int32_t frow = L.ctl[KAS_CTL_FAILROW];
    if (hung || frow >= 0) {                                 // (workgroup-uniform)
      failed = true; fail_topic = k;
      fail_part = hung ? -1 : (T.pid_arr ? T.pid_arr[frow] : frow);
      moved_r -= tr0.moved_replicas; moved_p -= tr0.moved_partitions;
      if constexpr (FS) {
        // the order wavefront may have emitted rows of this topic: it stops when it sees fs[1], says so in fs[2], and only
        // then is the topic padded (bounded like every wait between wavefronts: kas_solver_body.h, "Hang containment")
        if (lane == 0) kasw::store_shared_u64_lds(&fs[1], (uint64_t)(uint32_t)k);
        int32_t idle = 0;
        for (;;) {
          kasw::repoll();
          if (kasw::ballot(kasw::load_shared_u64_lds(&fs[2]) != 0ull) != 0ull) break;
          if (watchdog_poll(reinterpret_cast<uint32_t*>(&fs[3]), false, idle)) break;
        }
      }
      out_pad(topic_out(a, td), (int64_t)td.n_partitions * td.out_width, tid, NT);   // nothing is returned for a failed topic
      if (tid == 0) {
        kas_topic_result tr;
        tr.status = hung ? KAS_FAIL_WATCHDOG : KAS_FAIL_UNASSIGNABLE; tr.fail_partition = fail_part;
        tr.moved_replicas = 0; tr.moved_partitions = 0;
        a.topic_results[ti] = tr;
        kas_scenario_result sr;
        sr.status = tr.status; sr.fail_topic = fail_topic; sr.fail_partition = fail_part;
        sr.moved_replicas = moved_r; sr.moved_partitions = moved_p; sr.reserved = 0; sr.digest = 0;
        a.scenario_results[s] = sr;
      }
    }
  }
  if constexpr (FS) {
    kasw::wave_sync();                                       // (the records above before the word)
    if (lane == 0) kasw::store_shared_u64_lds(&fs[0], (uint64_t)(uint32_t)sd.topic_count << 32);
  }
  if (tid == 0 && a.stats) a.stats[(int64_t)s * KAS_STATS_PER_SCENARIO + 3] += kasw::clock_ticks() - t_begin;
}

// ---------------------------------------------------------------------------------------------
// Spread fill.  One workgroup streams a 1M-row scenario in ~10 ms (four wavefronts, each waiting on
// its own HBM round trips); the two row scans of the rack-diverse fill only meet at the quota, so for
// batches of few large single-topic scenarios they run over `sp_chunks` one-wavefront workgroups per
// scenario, one contiguous chunk of tiles each, as kernels of their own:
//   A   (scenario, chunk): sweep histogram of the chunk -> sp_hist; not rack-diverse / anything the
//       path does not cover -> sp_flag (the one-workgroup kernel then takes that scenario)
//   Q   (scenario, node): totals -> r*, quota, load (fill_quota's arithmetic) -> sp_node; the quota
//       left when chunk c starts -> sp_quota (prefix over the chunks' sweep-r* counts)
//   B   (scenario, chunk): pass B + P3 of the chunk (mid rows, orphan list at the chunk's place in
//       the scenario's list region, movement counts)
//   P4  (scenario): the chunk lists moved together, then first fit and the result records exactly as
//       in fill_topic (p4_lists_parallel on four wavefronts)
// Same row scans, same quota arithmetic, same P4: only where they run differs.
// ---------------------------------------------------------------------------------------------
struct SpreadTopic {
  TopicView T;
  NodeMap nm;
  kas_topic_desc td;
  int32_t ti;
  bool ok;              // the spread path covers this scenario
};

// what every phase derives first (uniform over the workgroup); LDS tables are not touched
template <int W>
KAS_DEV SpreadTopic spread_topic(const KasLaunch& a, int32_t s) {
  SpreadTopic S;
  const kas_scenario_desc sd = a.scen[s];
  const int32_t N = sd.n_nodes;
  S.ok = sd.topic_count == 1 && N > 0 && sd.ctx_off < 0;
  S.ti = sd.topic_begin;
  S.td = a.topics[S.ok ? sd.topic_begin : 0];
  const kas_topic_desc& td = S.td;
  TopicView& T = S.T;
  T.c16 = false;                                          // (the spread fill is not launched for 16-bit cells)
  T.cur = a.cur + td.cur_off;
  T.orph = a.orph + a.orph_off[s];
  T.len_arr = nullptr; T.inp_arr = nullptr;
  T.pid_arr = td.part_id_off >= 0 ? a.aux + td.part_id_off : nullptr;
  T.P = td.n_partitions; T.cw = td.cur_width; T.rf = td.rf; T.ow = td.out_width;
  T.hash = td.name_hash; T.nt = (T.P + 63) >> 6; T.N = N;
  T.mid = mid_base(a.out + td.out_off, T.P, T.ow);
  T.cap = S.ok ? max_replicas_per_node(N, T.P, T.rf) : 0;
  // rows exactly W wide without length / membership arrays (what the full-row stream reads), the
  // reference's pre-checks passed, a quota word that holds cap, a valid rotation (KAS:190)
  S.ok = S.ok && td.cur_width == W && td.out_width == W && td.cur_len_off < 0 && td.in_partitions_off < 0 &&
         td.rf > 0 && td.rf <= N && T.P > 0 && T.cap >= 0 && T.cap < (1 << 28) && java_abs_mod(td.name_hash, N) >= 0 &&
         !(a.flags & KAS_FLAG_GENERIC_FILL);
  S.nm.n = N; S.nm.min_id = 0; S.nm.range = 0u;
  if (N > 0) {
    const int64_t lo = a.node_id[sd.node_off], hi = a.node_id[sd.node_off + N - 1];
    const int64_t range = hi - lo + 1;
    S.nm.min_id = (int32_t)lo;
    if (range >= 1 && range <= (int64_t)a.idmap_entries) S.nm.range = (uint32_t)range;
  }
  S.ok = S.ok && S.nm.range != 0u;            // (sparse ids: the one-workgroup kernel's binary search)
  return S;
}

// mode 0: the layout of the one-workgroup fill (spread_p4); 1, 2: the slim layouts of the scans (kas_spread_scan_lds)
KAS_DEV LdsView spread_lds(const KasLaunch& a, unsigned char* lds_raw, int W, int NW, int mode = 0) {
  const KasLds lay = mode == 0 ? kas_fill_lds_layout(a.n_max, W, NW, a.idmap_entries, a.need_bsearch, 1)
                               : kas_spread_scan_lds(a.n_max, W, a.idmap_entries, a.need_bsearch, mode);
  LdsView L;
  L.x = (int32_t*)(lds_raw + lay.off_x);
  L.load = (int32_t*)(lds_raw + lay.off_load);
  L.qrs = (int32_t*)(lds_raw + lay.off_qrs);
  L.rack = (int16_t*)(lds_raw + lay.off_rack);
  L.live = (int16_t*)(lds_raw + lay.off_live);
  L.ns = 1; L.rs = 1;
  L.idmap = (int16_t*)(lds_raw + lay.off_idmap);
  L.ids = (int32_t*)(lds_raw + lay.off_ids);
  L.ring_p = (int32_t*)(lds_raw + lay.off_ring);
  L.ring_meta = L.ring_p + KAS_RING_CAP;
  L.ring_rack = (int16_t*)(L.ring_meta + KAS_RING_CAP);
  L.ctl = (int32_t*)(lds_raw + lay.off_ctl);
  return L;
}

// rack[] and the broker id -> node index table of a scenario (every thread of the workgroup; two barriers)
KAS_DEV void spread_node_tables(const KasLaunch& a, int32_t s, const LdsView& L, const NodeMap& nm, int32_t nt_threads) {
  const int tid = kasw::tid();
  const kas_scenario_desc sd = a.scen[s];
  const int32_t* g_node_id = a.node_id + sd.node_off;
  const int32_t* g_node_rack = a.node_rack + sd.node_off;
  for (int32_t i = tid; i < nm.n; i += nt_threads) lds_rack(L, i) = (int16_t)g_node_rack[i];
  for (uint32_t i = (uint32_t)tid; i < nm.range; i += (uint32_t)nt_threads) L.idmap[i] = (int16_t)-1;
  kasw::sync();
  for (int32_t i = tid; i < nm.n; i += nt_threads) L.idmap[(uint32_t)g_node_id[i] - (uint32_t)nm.min_id] = (int16_t)i;
  kasw::sync();
}

// node table checks of fill_scenario (strictly ascending non-negative ids, racks in int16 range)
KAS_DEV bool spread_nodes_bad(const KasLaunch& a, int32_t s) {
  const kas_scenario_desc sd = a.scen[s];
  bool bad = false;
  for (int32_t i = kasw::lane(); i < sd.n_nodes; i += 64) {
    const int32_t id = a.node_id[sd.node_off + i];
    const int32_t prev = i > 0 ? a.node_id[sd.node_off + i - 1] : -1;
    const int32_t rk = a.node_rack[sd.node_off + i];
    bad = bad || id <= prev || rk < 0 || rk > 32767;
  }
  return kasw::ballot(bad) != 0ull;
}

// phase A: one wavefront, chunk c of scenario s
template <int W>
KAS_DEV void spread_pass_a(const KasLaunch& a, int32_t s, int32_t c, unsigned char* lds_raw) {
  const int lane = kasw::lane();
  const int32_t CH = a.sp_chunks;
  SpreadTopic S = spread_topic<W>(a, s);
  if (S.ok && spread_nodes_bad(a, s)) S.ok = false;               // (before the id table is built from it)
  if (!S.ok) {
    if (lane == 0) a.sp_flag[s] = 1;
    return;
  }
  const LdsView L = spread_lds(a, lds_raw, W, 1, 1);
  const TopicView& T = S.T;
  const int32_t N = T.N;
  spread_node_tables(a, s, L, S.nm, 64);
  for (int32_t i = lane; i < (N * W + 1) / 2; i += 64) L.x[i] = 0;   // uint16 cells
  kasw::sync();
  const int32_t t0 = (int32_t)(((int64_t)T.nt * c) / CH), t1 = (int32_t)(((int64_t)T.nt * (c + 1)) / CH);
  const bool viol = fill_pass_a_range<W, true, true>(L, T, S.nm, t0, 1, t1);
  if (kasw::ballot(viol) != 0ull && lane == 0) a.sp_flag[s] = 1;   // not rack-diverse: the general fill's case
  kasw::sync();
  int32_t* g = a.sp_hist + ((int64_t)s * CH + c) * W * a.n_max;
  const uint16_t* h = (const uint16_t*)L.x;
  for (int32_t r = 0; r < W; ++r)
    for (int32_t n = lane; n < N; n += 64) g[(int64_t)r * a.n_max + n] = (int32_t)h[r * N + n];
}

// phase Q: node n of scenario s (no LDS; any launch shape)
template <int W>
KAS_DEV void spread_quota(const KasLaunch& a, int32_t s, int32_t n) {
  if (a.sp_flag[s] != 0) return;
  const kas_scenario_desc sd = a.scen[s];
  if (n >= sd.n_nodes) return;
  const kas_topic_desc td = a.topics[sd.topic_begin];
  const int32_t cap = max_replicas_per_node(sd.n_nodes, td.n_partitions, td.rf);
  const int32_t CH = a.sp_chunks;
  const int32_t* g = a.sp_hist + (int64_t)s * CH * W * a.n_max;
  int32_t tot[W];
#pragma unroll
  for (int r = 0; r < W; ++r) tot[r] = 0;
  for (int32_t c = 0; c < CH; ++c)
#pragma unroll
    for (int r = 0; r < W; ++r) tot[r] += g[((int64_t)c * W + r) * a.n_max + n];
  int32_t cum = 0, rs = W, q = 0;                               // as fill_quota
#pragma unroll
  for (int r = 0; r < W; ++r) {
    const int32_t cnt = tot[r];
    const bool sat = rs == W && cnt > cap - cum;
    q = sat ? cap - cum : q;
    cum = rs == W ? (sat ? cap : cum + cnt) : cum;
    rs = sat ? r : rs;
  }
  a.sp_node[((int64_t)s * 2 + 0) * a.n_max + n] = cum;
  a.sp_node[((int64_t)s * 2 + 1) * a.n_max + n] = (int32_t)(((uint32_t)rs << 28) | (uint32_t)q);
  int32_t rem = q;
  for (int32_t c = 0; c < CH; ++c) {
    a.sp_quota[((int64_t)s * CH + c) * a.n_max + n] = rem;
    rem -= rs < W ? g[((int64_t)c * W + rs) * a.n_max + n] : 0;
  }
}

// phase B: one wavefront, chunk c of scenario s
template <int W>
KAS_DEV void spread_pass_b(const KasLaunch& a, int32_t s, int32_t c, unsigned char* lds_raw) {
  const int lane = kasw::lane();
  if (a.sp_flag[s] != 0) return;                                 // (written by an earlier kernel: uniform)
  const int32_t CH = a.sp_chunks;
  SpreadTopic S = spread_topic<W>(a, s);
  const LdsView L = spread_lds(a, lds_raw, W, 1, 2);
  const TopicView& T = S.T;
  const int32_t N = T.N;
  spread_node_tables(a, s, L, S.nm, 64);
  for (int32_t n = lane; n < N; n += 64) {
    lds_qrs(L, n) = a.sp_node[((int64_t)s * 2 + 1) * a.n_max + n];
    L.x[n] = a.sp_quota[((int64_t)s * CH + c) * a.n_max + n];
  }
  kasw::sync();
  const int32_t t0 = (int32_t)(((int64_t)T.nt * c) / CH), t1 = (int32_t)(((int64_t)T.nt * (c + 1)) / CH);
  int32_t moved_r = 0, moved_p = 0;
  int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int32_t oc = fill_pass_b_range<W, true, 1>(L, T, S.nm, t0, t1, L.x, moved_r, moved_p, st);
  const int32_t mr = kasw::wave_sum(moved_r), mp = kasw::wave_sum(moved_p);
  if (lane == 0) {
    int32_t* g = a.sp_oc + (int64_t)s * (CH + 2);
    g[c] = oc;
    if (mr != 0) kasw::global_atomic_add(&g[CH], mr);
    if (mp != 0) kasw::global_atomic_add(&g[CH + 1], mp);
  }
}

// phase P4: NW wavefronts, scenario s
template <int W, int NW>
KAS_DEV void spread_p4(const KasLaunch& a, int32_t s, unsigned char* lds_raw) {
  constexpr int NT = 64 * NW;
  const int lane = kasw::lane();
  const int tid = kasw::tid();
  const int32_t wave = kasw::wave_id();
  if (a.sp_flag[s] != 0) return;
  const int32_t CH = a.sp_chunks;
  SpreadTopic S = spread_topic<W>(a, s);
  const LdsView L = spread_lds(a, lds_raw, W, NW, 3);
  TopicView& T = S.T;
  const int32_t N = T.N, cap = T.cap;
  const int64_t t_begin = kasw::clock_ticks();
  for (int32_t i = tid; i < N; i += NT) {
    lds_load(L, i) = a.sp_node[((int64_t)s * 2 + 0) * a.n_max + i];
    lds_rack(L, i) = (int16_t)a.node_rack[a.scen[s].node_off + i];
  }
  if (tid < KAS_CTL_INTS) L.ctl[tid] = tid == KAS_CTL_FAILROW ? -1 : (tid == KAS_CTL_FAILWIN ? 0x7fffffff : 0);
  kasw::sync();
  // the chunks' orphan lists, moved together at the start of the scenario's list region (ascending, so
  // a list only ever moves down; 64 * NW entries at a time: read, barrier, write, barrier)
  const int32_t* oc = a.sp_oc + (int64_t)s * (CH + 2);
  int32_t total = 0;
  for (int32_t c = 0; c < CH; ++c) {
    const int32_t len = oc[c];
    const int64_t src = ((int64_t)(((int64_t)T.nt * c) / CH)) << 6;
    if ((int64_t)total != src) {
      for (int32_t i0 = 0; i0 < len; i0 += NT) {
        const int32_t v = i0 + tid < len ? T.orph[src + i0 + tid] : 0;
        kasw::sync();
        if (i0 + tid < len) T.orph[(int64_t)total + i0 + tid] = v;
        kasw::sync();
      }
    }
    total += len;
  }
  if (tid == 0) L.ctl[KAS_CTL_OC] = total;                      // one list: chunk 0 of p4_lists_parallel holds them all
  // KAS:168 getNodeProcessingOrder + the non-full nodes in that order, as in fill_topic
  const int32_t idxN = java_abs_mod(T.hash, N);
  if (wave == 0) {
    const int32_t start = (N - idxN) % N;
    int32_t live_count = 0;
    for (int32_t base = 0; base < N; base += 64) {
      const int32_t j = base + lane;
      int32_t n = j + start; if (n >= N) n -= N;
      const bool is_live = j < N && lds_load(L, n) < cap;
      const uint64_t m = kasw::ballot(is_live);
      if (is_live) L.live[live_count + kasw::count_below(m)] = (int16_t)n;
      live_count += kasw::popc(m);
    }
    kasw::lockstep();
    if (lane == 0) L.ctl[KAS_CTL_LIVE] = live_count;
  }
  kasw::sync();
  int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int32_t fail_win = -1, fail_row = -1;
  p4_lists_parallel<W, NW>(L, T, L.ctl[KAS_CTL_LIVE], wave, st, fail_win, fail_row);
  kasw::sync();
  if (fail_win >= 0 && fail_win == L.ctl[KAS_CTL_FAILWIN] && lane == 0) L.ctl[KAS_CTL_FAILROW] = fail_row;
  kasw::sync();
  TopicOutcome o;
  o.status = KAS_OK; o.fail_partition = -1; o.moved_replicas = oc[CH]; o.moved_partitions = oc[CH + 1];
  if (KAS_SPIN_BOUND > 0 && L.ctl[KAS_CTL_WATCHDOG] != 0) o.status = KAS_FAIL_WATCHDOG;
  else if (L.ctl[KAS_CTL_FAILROW] >= 0) {                       // KAS:183-184
    o.status = KAS_FAIL_UNASSIGNABLE;
    o.fail_partition = T.pid_arr ? T.pid_arr[L.ctl[KAS_CTL_FAILROW]] : L.ctl[KAS_CTL_FAILROW];
  }
  if (o.status != KAS_OK) {                                     // nothing is returned for a failed topic
    int32_t* out = a.out + S.td.out_off;
    const int64_t cells = (int64_t)S.td.n_partitions * S.td.out_width;
    for (int64_t i = tid; i < cells; i += NT) out[i] = -1;
    o.moved_replicas = 0; o.moved_partitions = 0;
  }
  if (tid == 0) {
    kas_topic_result tr;
    tr.status = o.status; tr.fail_partition = o.fail_partition;
    tr.moved_replicas = o.moved_replicas; tr.moved_partitions = o.moved_partitions;
    a.topic_results[S.ti] = tr;
    kas_scenario_result sr;
    sr.status = o.status; sr.fail_topic = o.status != KAS_OK ? 0 : -1; sr.fail_partition = o.fail_partition;
    sr.moved_replicas = o.moved_replicas; sr.moved_partitions = o.moved_partitions; sr.reserved = 0;
    sr.digest = 0;
    a.scenario_results[s] = sr;
    if (a.stats) {
      for (int i = 0; i < 8; ++i) a.stats[(int64_t)s * KAS_STATS_PER_SCENARIO + i] = 0;
      a.stats[(int64_t)s * KAS_STATS_PER_SCENARIO + 3] = kasw::clock_ticks() - t_begin;   // [3] P4 (with the list moves)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Between fill and order: perm[] = scenario indices by descending number of orphans placed (the
// length of a scenario's P5 dependency chain, known from the fill kernel's movement count).  The
// order kernel takes scenarios in this order, so the G scenarios sharing a solver wavefront have
// chains of similar length (the wavefront lasts as long as its longest chain) and the longest
// ones start first.  One workgroup of KAS_PERM_WAVES wavefronts, any batch size: a counting sort
// over KAS_PERM_BINS key classes (key >> shift, largest class first; inside a class the order is
// whatever the atomics make it — scenarios of one class are as good as equal for pairing, and no
// result depends on the permutation).  LDS: int32 bins[KAS_PERM_BINS + 1] + one word.
// ---------------------------------------------------------------------------------------------
#define KAS_PERM_WAVES 4
#define KAS_PERM_BINS 4096
KAS_DEV void order_permutation(const KasLaunch& a, unsigned char* lds_raw) {
  constexpr int NT = 64 * KAS_PERM_WAVES;
  const int32_t S = a.n_scenarios;
  const int tid = kasw::tid();
  int32_t* bins = (int32_t*)lds_raw;
  int32_t* kmax = bins + KAS_PERM_BINS + 1;
  for (int32_t i = tid; i <= KAS_PERM_BINS; i += NT) bins[i] = 0;
  if (tid == 0) *kmax = 0;
  kasw::sync();
  int32_t m = 0;
  for (int32_t i = tid; i < S; i += NT) {
    const int32_t k = a.scenario_results[i].moved_replicas;
    m = k > m ? k : m;
  }
  kasw::lds_atomic_max(kmax, m);
  kasw::sync();
  int32_t shift = 0;
  while (((*kmax) >> shift) >= KAS_PERM_BINS) ++shift;      // workgroup-uniform
  for (int32_t i = tid; i < S; i += NT) {
    const int32_t k = a.scenario_results[i].moved_replicas;
    const int32_t cls = KAS_PERM_BINS - 1 - ((k > 0 ? k : 0) >> shift);   // largest keys first
    kasw::lds_atomic_add(&bins[cls + 1], 1);
  }
  kasw::sync();
  // inclusive prefix over the classes (bins[c + 1] = scenarios in classes <= c): each thread scans
  // its own run of classes, one thread adds up the run totals, every thread adds its run's offset
  constexpr int RUN = KAS_PERM_BINS / NT;
  int32_t run_sum = 0;
  for (int32_t c = 0; c < RUN; ++c) { run_sum += bins[1 + tid * RUN + c]; bins[1 + tid * RUN + c] = run_sum; }
  kasw::sync();
  if (tid == 0) {
    int32_t acc = 0;
    for (int32_t t = 0; t < NT; ++t) { const int32_t v = bins[1 + t * RUN + RUN - 1]; bins[1 + t * RUN + RUN - 1] = acc + v; acc += v; }
  }
  kasw::sync();
  // (the last class of every run now holds the global inclusive count; the others still lack the
  // offset of the runs before theirs)
  const int32_t before = tid > 0 ? bins[1 + tid * RUN - 1] : 0;
  for (int32_t c = 0; c + 1 < RUN; ++c) bins[1 + tid * RUN + c] += before;
  kasw::sync();
  // bins[c] = first position of class c (bins[0] = 0); hand out positions
  for (int32_t i = tid; i < S; i += NT) {
    const int32_t k = a.scenario_results[i].moved_replicas;
    const int32_t cls = KAS_PERM_BINS - 1 - ((k > 0 ? k : 0) >> shift);
    a.perm[kasw::lds_atomic_add(&bins[cls], 1)] = i;
  }
}

// ---------------------------------------------------------------------------------------------
// order kernel, ticket form.  A workgroup is three wavefronts serving G scenarios (one lane group
// of GL = 64 / G lanes each): wave 0 SOLVES, wave 1 STAGES rows for it, wave 2 RETIRES what it
// finished.  Lane l of every wave owns the rows li, li + GL, li + 2 GL, ... (li = l % GL) of every
// solved topic of its scenario, in order.
//
// The solver's loop is a chain of dependent steps (rows waiting for earlier rows on a shared
// node), so nothing with memory latency may sit in it: it touches LDS only.  The stager streams the fill kernel's
// rows (node indices) from HBM one GL-row tile at a time, hands out tickets in row order
// (ticket of (row, node) = how many earlier rows of the scenario hold that node: a per-node
// running count + the rank among the tile's lanes holding it, from one lane mask per node),
// and stages each row into a ring of KAS_RING_SLOTS 16-byte slots per lane; the retirer turns
// finished rows into the final out row (node index -> broker id) and the digest.  Slot protocol
// (tag = first dword):
//     FREE  --stager-->  j (= the lane's j-th row is staged)  --solver-->  DONE | picks
//     --retirer-->  FREE                                      END = the lane has no more rows
//
// count[n][r] lives in LDS as 4 x uint16 per node, the fourth field counting the rows that
// committed on the node.  A row commits once "commits on n == its ticket" holds for each of its
// nodes (every earlier row holding n has committed: KAS:225-236 runs rows in ascending order),
// with one LDS atomic add per node.  Lanes never wait for each other except through tickets, so
// the scenarios of one wavefront do not interact at all.
// ---------------------------------------------------------------------------------------------
#ifndef KAS_RING_SLOTS
#define KAS_RING_SLOTS 4
#endif
// rows a run must decide beyond the ones that were ready anyway for its path to pay
// a row waiting on exactly one node with this many rows ahead of it nominates the node
#ifndef KAS_RUN_NOMINATE
#define KAS_RUN_NOMINATE 2
#endif
// s_sleep argument of a stager / retirer iteration that found nothing to do (~64 cycles each)
#ifndef KAS_IDLE_NAP
#define KAS_IDLE_NAP 4
#endif
#ifndef KAS_SOLVER_PRIO
#define KAS_SOLVER_PRIO 3
#endif
// retiring wave: rows per lane whose id reads are in flight together (two such batches alternate)
#ifndef KAS_RETIRE_UR
#define KAS_RETIRE_UR 2
#endif
// diagnostics build: rows in hand / rows ready per solver step, iterations of the retiring wave (kas_plan_stats [15], [7], [4], [5])
#ifndef KAS_ORDER_DIAG
#define KAS_ORDER_DIAG 0
#endif
// solver: lanes left without a row after the claim (of 64) from which on the wave naps (s_sleep argument; 0 = never)
#ifndef KAS_STARVE_NAP
#define KAS_STARVE_NAP 0
#endif
#ifndef KAS_STARVE_MIN
#define KAS_STARVE_MIN 16
#endif
#ifndef KAS_RUN_MIN_GAIN
#define KAS_RUN_MIN_GAIN 3
#endif
// a queue pass that gained less than that makes the next 1, 2, 4 ... KAS_RUN_BACKOFF_MAX nominations be skipped
#ifndef KAS_RUN_BACKOFF_MAX
#define KAS_RUN_BACKOFF_MAX 16
#endif
#define KAS_TAG_FREE (-1)
#define KAS_TAG_END  (-3)
#define KAS_TAG_DONE ((int32_t)0x80000000)   // | w0 | w1 << 2 | Lp << 4
#define KAS_TAG_IS_DONE(t) (((uint32_t)(t) & 0xffffff00u) == 0x80000000u)
// staged tag: the lane's row counter j in bits 0..25; bits 26..28 = for each stored position w0
// that the first pick may take, whether the HIGHER of the two remaining stored positions is
// visited first by the second pick; bits 29..30 = list length Lp (1..3)
#define KAS_TAG_JMASK 0x03ffffff
#define KAS_DUMMY_COUNTS 0x0000ffffffffffffull   // counter row of the padding holder: never picked

struct alignas(16) RingSlot { int32_t tag; int32_t c[3]; };

// a lane's tile sequence: GL-row tiles of every topic the fill kernel solved, in order
struct TileIter {
  int32_t k, tP, tow, row0;              // topic, its rows / row width, first row of the current tile
  int32_t rot;                           // the topic's rotation word: ticket_rotation() (lists <= 3 wide)
                                         // or idx_m at bit 3 m for m = 1..5 (wide lists)
  int64_t tout;
  int64_t tmid;                          // first mid row of the topic: uint16 offset from the out pool
  bool exhausted;
};

// How a staged row is laid out for a topic (KAS:190, 263-278).  The holders of a row, ascending,
// are visited by a pick over m of them starting at offset idx_m = abs(hash) % m.  Returned word:
//   bits 0..5   for m = 3: which ascending holder goes to stored position 0, 1, 2 (2 bits each) so
//               that stored order == the FIRST pick's visit order
//   bits 6..11  the same for m = 2 (the third stored position keeps the padding holder)
//   bits 12..14 for m = 3 and each stored position w0 the first pick may take: whether the HIGHER
//               of the two remaining stored positions is visited first by the second pick
KAS_DEV int32_t ticket_rotation(int32_t name_hash) {
  const int32_t idx2 = java_abs_mod(name_hash, 2), idx3 = java_abs_mod(name_hash, 3);
  int32_t rank3[3], word = 0;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    int32_t r = t + 3 - idx3;                               // rank visited at position t: (t + m - idx_m) % m
    r -= r >= 3 ? 3 : 0;
    rank3[t] = r;
    word |= r << (2 * t);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    int32_t r = t + 2 - idx2;
    r -= r >= 2 ? 2 : 0;
    word |= r << (6 + 2 * t);
  }
  word |= 2 << 10;
#pragma unroll
  for (int w0 = 0; w0 < 3; ++w0) {
    // second pick: the two remaining holders in RANK order are visited (lower, higher) when
    // idx2 == 0 and (higher, lower) when idx2 == 1
    const int pp = w0 == 0 ? 1 : 0, qq = w0 == 2 ? 1 : 2;  // remaining stored positions
    const bool q_lower_rank = rank3[qq] < rank3[pp];
    const bool q_first = idx2 == 0 ? q_lower_rank : !q_lower_rank;
    word |= q_first ? (1 << (12 + w0)) : 0;
  }
  return word;
}
#define KAS_ROT_IDENT (0 | (1 << 2) | (2 << 4))

// next topic the fill kernel solved (rarely taken: kept out of the tile loops; by value so that
// the iterator stays in registers)
template <bool WIDE>
KAS_DEV_COLD TileIter tile_next_topic(TileIter it, const KasLaunch& a, const kas_scenario_desc& sd) {
  for (;;) {
    it.k += 1;
    if (it.k >= sd.topic_count) { it.exhausted = true; return it; }
    const int32_t ti = sd.topic_begin + it.k;
    if (a.topic_results[ti].status != KAS_OK) continue;
    const kas_topic_desc td = a.topics[ti];
    if (td.n_partitions <= 0) continue;
    it.tP = td.n_partitions; it.tow = td.out_width; it.tout = td.out_off;
    it.tmid = 2 * (td.out_off + (int64_t)it.tP * it.tow) - (int64_t)it.tP * mid_width(it.tow);
    if (WIDE) {
      it.rot = 0;
#pragma unroll
      for (int m = 1; m <= 5; ++m) it.rot |= (java_abs_mod(td.name_hash, m) & 7) << (3 * m);
    } else {
      it.rot = ticket_rotation(td.name_hash);
    }
    it.row0 = 0;
    return it;
  }
}

template <int GL, bool WIDE = false>
KAS_DEV bool tile_next(TileIter& it, const KasLaunch& a, const kas_scenario_desc& sd) {
  if (it.exhausted) return false;
  it.row0 += GL;
  if (it.row0 < it.tP) return true;
  it = tile_next_topic<WIDE>(it, a, sd);
  return !it.exhausted;
}

// first mid row of the iterator's topic, as a uint16 offset from the out pool
KAS_DEV int64_t tile_mid_offset(const TileIter& it) { return it.tmid; }

KAS_DEV TileIter tile_iter_begin(bool have_scenario) {
  TileIter it;
  it.k = -1; it.tP = 0; it.tow = 1; it.row0 = 0; it.rot = KAS_ROT_IDENT; it.tout = 0; it.tmid = 0;
  it.exhausted = !have_scenario;
  return it;
}

// The solver's two picks over a staged row (stored positions 0..2 = the first pick's visit order).
// First pick: count[.][0], first strictly smaller wins == the minimum of (count, stored position).
// Second pick: count[.][1] of the two remaining stored positions; the one visited first wins ties:
// minimum of (count, visited second, stored position), the taken one excluded.  The third is what
// is left.
KAS_DEV void ticket_picks(const uint32_t (&f0)[3], const uint32_t (&f1)[3], int32_t meta, int32_t& w0, int32_t& w1) {
  const uint32_t k0 = f0[0] << 2, k1 = (f0[1] << 2) | 1u, k2 = (f0[2] << 2) | 2u;
  const uint32_t kmin = k0 < k1 ? (k0 < k2 ? k0 : k2) : (k1 < k2 ? k1 : k2);
  w0 = (int32_t)(kmin & 3u);
  const uint32_t vis = (uint32_t)meta >> w0;                // bit 0: the higher remaining position is visited first
  const uint32_t hi_first = vis & 1u;
  const uint32_t lo_pos = w0 == 0 ? 1u : 0u, hi_pos = w0 == 2 ? 1u : 2u;
  const uint32_t c_lo = w0 == 0 ? f1[1] : f1[0], c_hi = w0 == 2 ? f1[1] : f1[2];
  const uint32_t key_lo = (c_lo << 3) | (hi_first << 2) | lo_pos;
  const uint32_t key_hi = (c_hi << 3) | ((hi_first ^ 1u) << 2) | hi_pos;
  w1 = (int32_t)((key_lo < key_hi ? key_lo : key_hi) & 3u);
}

// Context handed in (KAS:360-369) and the ticket forms.  Tickets and commit counts start at zero in every
// solve, so a Context only seeds the count fields of the counter rows (and is read back from them at
// the end); what has to hold is "largest counter + rows the node can gain in this solve < field limit".
// ctx_gain_bound = the second term: the sum of the topics' caps (KAS:65-71), the same bound the plan
// keeps below the ticket limits (kas_shape_batch).  A scenario that fails the test is left to the round
// form (KasLaunch::ord_flag, KAS_FLAG_ORDER_FLAGGED): every wavefront of the workgroup evaluates the
// test over the whole table by itself, so they agree without a word of LDS.
KAS_DEV uint32_t ctx_gain_bound(const KasLaunch& a, const kas_scenario_desc& sd) {
  int64_t b = 0;
  for (int32_t k = 0; k < sd.topic_count; ++k) {
    const kas_topic_desc td = a.topics[sd.topic_begin + k];
    if (td.rf >= 1 && td.rf <= sd.n_nodes) b += ((int64_t)td.n_partitions * td.rf + sd.n_nodes - 1) / sd.n_nodes;
  }
  return b > 0x7fffffff ? 0x7fffffffu : (uint32_t)b;
}
// some counter of columns [0, cols) of rows li, li + stride, ... does not leave room for `gain` more below `limit`
KAS_DEV bool ctx_over_limit(const int32_t* g_ctx, int32_t N, int32_t ctxw, int32_t cols, uint32_t limit, uint32_t gain,
                            int32_t li, int32_t stride) {
  bool over = false;
  const uint32_t room = limit > gain ? limit - gain : 0u;   // counters must stay below this (negative ones read as huge)
  for (int32_t n = li; n < N; n += stride)
    for (int32_t r = 0; r < cols; ++r) over = over || (uint32_t)g_ctx[(int64_t)n * ctxw + r] >= room;
  return over;
}

// PK: counter rows are one uint32 of three 10-bit counts (commits on the node = their sum) instead
// of 4 x uint16 — half the LDS, for scenarios whose per-node row count stays below 1023.
template <int W, int G, bool PK>
KAS_DEV void order_tickets(const KasLaunch& a, int32_t first_scenario, unsigned char* lds_raw) {
  static_assert(W <= 3, "ring slots and packed counter rows hold lists up to 3 wide");
  constexpr int RB = PK ? 4 : 8;                            // bytes per counter row
  constexpr int GL = 64 / G;
  constexpr int HALVES = GL > 32 ? 2 : 1;                   // ticket pass: 32 rows per lane mask
  constexpr int K = KAS_RING_SLOTS;
  const int lane = kasw::lane();
  const int32_t wave = kasw::wave_id();
  const int32_t g = lane / GL, li = lane % GL;
  const bool have_s0 = first_scenario + g < a.n_scenarios;
  const int32_t s = have_s0 ? (a.perm ? a.perm[first_scenario + g] : first_scenario + g) : a.n_scenarios;
  const int32_t nmax = a.n_max > 0 ? a.n_max : 1;
  const int32_t cnt_base = g * kas_order_ticket_group_bytes(a.n_max, G, PK);   // LDS byte offset of this group's region
  unsigned char* cnt = lds_raw + cnt_base;                  // [nmax + 1] rows: + the padding holder's row
  uint32_t* dep = (uint32_t*)(lds_raw + cnt_base + kas_align16(RB * (int64_t)(nmax + 1)));   // lane mask per node (ticket pass)
  uint16_t* run = (uint16_t*)(dep + nmax + 1);              // tickets handed out per node so far (both: + the padding node)
  // padding holder: counts that never win a pick, and a ticket that always matches its commits
  const int32_t dummy_addr = PK ? ((3 * 0x3ff) << 16) | (cnt_base + nmax * RB) : (cnt_base + nmax * RB);
  RingSlot* ring = (RingSlot*)(lds_raw + G * kas_order_ticket_group_bytes(a.n_max, G, PK));
  uint64_t* gdig = (uint64_t*)(ring + K * 64);
  uint32_t* rank_owner = (uint32_t*)(gdig + G);             // [64] run scratch of the solver: rank -> lane
  uint32_t* wd = rank_owner + 64;                           // watchdog word (see watchdog_poll)

  kas_scenario_desc sd;
  sd.n_nodes = 0; sd.topic_begin = 0; sd.topic_count = 0; sd.ctx_width = 0; sd.node_off = 0; sd.ctx_off = -1;
  if (have_s0) sd = a.scen[s];
  const int32_t N = sd.n_nodes;
  const int32_t* g_node_id = a.node_id + sd.node_off;
  // Context handed in (4 x uint16 counter rows only: the plan never pairs a Context with the packed rows)
  bool have_s = have_s0;
  int32_t* g_ctx = nullptr;
  int32_t ccols = 0;
  if constexpr (!PK) {
    const bool ctx_s = have_s0 && sd.ctx_off >= 0 && sd.ctx_width > 0 && a.ctx != nullptr;
    if (kasw::ballot(ctx_s) != 0ull) {                       // (wave-uniform; no Context anywhere: nothing to do)
      constexpr uint64_t GLM = GL == 64 ? ~0ull : ((1ull << GL) - 1ull);
      bool over = false;
      if (ctx_s) {
        g_ctx = a.ctx + sd.ctx_off;
        ccols = sd.ctx_width < W ? sd.ctx_width : W;
        over = ctx_over_limit(g_ctx, N, sd.ctx_width, ccols, 65536u, ctx_gain_bound(a, sd), li, GL);
      }
      const bool flagged = ((kasw::ballot(over) >> (g * GL)) & GLM) != 0ull;   // my scenario goes to the round form
      if (flagged) { have_s = false; g_ctx = nullptr; ccols = 0; }
      if (flagged && wave == 0 && li == 0 && a.ord_flag) a.ord_flag[s] = 1;
    }
  }
  for (int32_t n = li + GL * wave; n < N; n += 3 * GL) {
    if (PK) ((uint32_t*)cnt)[n] = 0u;
    else {
      uint64_t x = 0ull;                                     // count[n][0..2] from the Context, commits 0
      for (int32_t r = 0; r < ccols; ++r) x |= (uint64_t)(uint32_t)g_ctx[(int64_t)n * sd.ctx_width + r] << (16 * r);
      ((uint64_t*)cnt)[n] = x;
    }
    run[n] = 0; dep[n] = 0u;
  }
  if (wave == 0 && li == 0) {
    if (PK) ((uint32_t*)cnt)[nmax] = 0x3fffffffu; else ((uint64_t*)cnt)[nmax] = KAS_DUMMY_COUNTS;
    gdig[g] = 0ull;
  }
  if (wave == 0) rank_owner[lane] = 0u;
  if (wave == 0 && lane == 0) *wd = 0u;
  for (int32_t k = wave; k < K; k += 3) ring[k * 64 + lane].tag = KAS_TAG_FREE;
  kasw::sync();
  int32_t wd_idle = 0;

  if (wave == 0) {
    // ------------------------------------------------------------------ solver: LDS only
    // cur = the row being decided, nxt = the lane's following row, read ahead from the ring so
    // that taking it costs no LDS round trip of its own
    // Rows are not tied to lanes: a lane without a row CLAIMS the next unclaimed row of its scenario
    // (rows in tile order: virtual row v = tile * GL + column lives in ring slot (tile % K, column)),
    // so the GL lanes of a group always hold the oldest rows that are still undecided — a lane
    // stuck behind a ticket no longer keeps the rows of "its" column from being worked on
    // (rows in hand 21 -> ~30 of 32).  Any lane may decide any row: the tickets order them.
    int32_t gnext = 0;                                       // next unclaimed virtual row (group-uniform)
    int32_t my_slot = lane;                                  // ring slot of the row in hand
    bool cv = false, gfin = false;
    int32_t e0 = dummy_addr, e1 = dummy_addr, e2 = dummy_addr, meta = 0;
    int64_t n_iter = 0, n_blocked = 0, n_relax = 0, n_run_rows = 0, n_runs = 0, n_cur = 0, n_rdy = 0;
    int32_t run_skip = 0, run_backoff = 0;                   // wave-uniform
    const int64_t t_begin = kasw::clock_ticks();
    kasw::set_priority<KAS_SOLVER_PRIO>();                 // the chain: first call on the SIMD's issue slots
    for (;;) {
      kasw::repoll();                                      // LDS is re-read below
      n_iter += 1;
      // one LDS round trip per iteration: the three counter rows (+ the look-ahead slot / the slot
      // of the row claimed at the end of the previous iteration)
      uint32_t f0[3], f1[3], com[3];                        // count[.][0], count[.][1], commits per holder
      const int32_t es[3] = {e0, e1, e2};
      if constexpr (PK) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const uint32_t x = *(const uint32_t*)(lds_raw + (es[q] & 0xffff));
          f0[q] = x & 0x3ffu; f1[q] = (x >> 10) & 0x3ffu;
          com[q] = f0[q] + f1[q] + ((x >> 20) & 0x3ffu);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const uint64_t x = *(const uint64_t*)(lds_raw + (es[q] & 0xffff));
          f0[q] = (uint32_t)x & 0xffffu; f1[q] = (uint32_t)x >> 16;
          com[q] = (uint32_t)(x >> 48);
        }
      }
      // rows still ahead of mine on each holder: ticket - commits on the node; 0 everywhere ==
      // every earlier row holding any of my nodes has committed
      const uint32_t d0 = ((uint32_t)e0 >> 16) - com[0], d1 = ((uint32_t)e1 >> 16) - com[1],
                     d2 = ((uint32_t)e2 >> 16) - com[2];
      const uint32_t d_any = d0 | d1 | d2, d_sum = d0 + d1 + d2;
      bool ready = cv && d_any == 0u;
      if ((KAS_COUNTERS_ON || KAS_ORDER_DIAG) && a.stats) n_cur += kasw::popc((kasw::ballot(cv) >> (g * GL)) & (GL == 64 ? ~0ull : ((1ull << GL) - 1ull)));
      // ---- runs.  First fit hands consecutive orphans to one node until it is full, so the rows
      // in hand often queue on ONE node X (tickets t, t+1, ...) while their other holders are
      // free.  Such a queue is decided in this iteration: its rows differ from "ready" only in
      // the counts of X they will see — X's counts now plus what the rows before them in the
      // queue add.  X wins a row's first pick iff count[X][0] is below a threshold fixed by the
      // row's other holders, else its second pick iff count[X][1] is below another; with the
      // queue laid out by rank (lane r of the group = the row r places behind X's commits) the
      // counts are prefix sums of those wins, and re-evaluating them until nothing changes gives
      // the sequential answer (row r is right after round r at the latest; usually 2 rounds).
      {
        // a row waiting on exactly one node with two rows ahead of it nominates that node
        // (queues that turn out too short to pay for this path — dense small clusters — make it
        // back off: the next attempts are skipped, twice as many each time, up to 16)
        uint64_t nb = 0ull;
        if (run_skip > 0) run_skip -= 1;
        else nb = kasw::ballot(cv && d_any == (uint32_t)KAS_RUN_NOMINATE && d_sum == (uint32_t)KAS_RUN_NOMINATE);
        if (nb != 0ull) {
          constexpr uint64_t GLM = GL == 64 ? ~0ull : ((1ull << GL) - 1ull);
          const int32_t gsh = g * GL;
          const int32_t a0 = e0 & 0xffff, a1 = e1 & 0xffff, a2 = e2 & 0xffff;
          const int32_t my_ax = d0 != 0u ? a0 : (d1 != 0u ? a1 : a2);
          int32_t ax = -1;                                    // my group's nominated node (its counter row)
#pragma unroll
          for (int gg = 0; gg < G; ++gg) {
            const uint64_t m = (nb >> (gg * GL)) & GLM;
            const int32_t v = kasw::read_lane(my_ax, m != 0ull ? gg * GL + kasw::first_lane(m) : 0);
            ax = (g == gg && m != 0ull) ? v : ax;
          }
          const int32_t hx = a0 == ax ? 0 : (a1 == ax ? 1 : (a2 == ax ? 2 : -1));
          const uint32_t kx = hx == 0 ? d0 : (hx == 1 ? d1 : d2);             // my rank in X's queue
          // candidates: rows in hand that hold X and wait for nothing else
          const bool cand = cv && hx >= 0 && d_sum == kx && kx < (uint32_t)GL;
          n_runs += 1;
          const uint32_t seq = (uint32_t)(n_runs & 0xffffff);          // never 0: stale and initial entries differ
          if (cand) rank_owner[gsh + (int32_t)kx] = (seq << 8) | (uint32_t)lane;
          kasw::lockstep();
          const uint32_t ow = rank_owner[lane];                // rank view: lane li of a group = rank li
          kasw::lockstep();
          const bool have = (ow >> 8) == seq;
          const uint64_t hb = (kasw::ballot(have) >> gsh) & GLM;
          const int32_t qlen = (~hb & GLM) != 0ull ? kasw::first_lane(~hb & GLM) : GL;   // ranks 0..qlen-1 are all in hand
          const bool member = cand && (int32_t)kx < qlen;
          const int32_t gain = kasw::popc(kasw::ballot(member && kx > 0u));   // rows beyond the ones ready anyway
          if (gain < KAS_RUN_MIN_GAIN) {
            run_backoff = run_backoff == 0 ? 1 : (run_backoff < KAS_RUN_BACKOFF_MAX ? 2 * run_backoff : KAS_RUN_BACKOFF_MAX);
            run_skip = KAS_RUN_BACKOFF_MAX > 0 ? run_backoff : 0;
          } else {
            if (gain > KAS_RUN_MIN_GAIN) run_backoff = 0;
            // thresholds of the owner's row (relative to X's counts now)
            const uint32_t k0 = f0[0] << 2, k1 = (f0[1] << 2) | 1u, k2 = (f0[2] << 2) | 2u;
            const uint32_t mo = hx == 0 ? (k1 < k2 ? k1 : k2) : (hx == 1 ? (k0 < k2 ? k0 : k2) : (k0 < k1 ? k0 : k1));
            const uint32_t bx0 = hx == 0 ? f0[0] : (hx == 1 ? f0[1] : f0[2]);
            const uint32_t bx1 = hx == 0 ? f1[0] : (hx == 1 ? f1[1] : f1[2]);
            // first pick: (c << 2 | hx) < mo  <=>  c < ceil((mo - hx) / 4)
            const int32_t t0 = (int32_t)((mo - (uint32_t)hx + 3u) >> 2) - (int32_t)bx0;
            // X loses the first pick to stored position o0; the second pick is between X and o1
            const int32_t o0 = (int32_t)(mo & 3u), o1 = 3 - hx - o0;
            const uint32_t hi_first = ((uint32_t)meta >> o0) & 1u;
            const uint32_t vx = hx > o1 ? hi_first ^ 1u : hi_first;          // "visited second" bit of X / of o1
            const uint32_t co1 = o1 == 0 ? f1[0] : (o1 == 1 ? f1[1] : f1[2]);
            const uint32_t ko = (co1 << 3) | ((vx ^ 1u) << 2) | (uint32_t)o1;
            const uint32_t lowx = (vx << 2) | (uint32_t)hx;
            // (c << 3 | lowx) < ko  <=>  c < ceil((ko - lowx) / 8)
            const int32_t t1 = (int32_t)((ko - lowx + 7u) >> 3) - (int32_t)bx1;
            const int32_t t0c = t0 < 0 ? 0 : (t0 > 127 ? 127 : t0), t1c = t1 < 0 ? 0 : (t1 > 127 ? 127 : t1);
            const int32_t th = kasw::shfl(t0c | (t1c << 8), have ? (int32_t)(ow & 0xffu) : lane);   // owner -> rank view
            const int32_t T0 = th & 0xff, T1 = (th >> 8) & 0xff;
            const bool act = li < qlen;
            const uint64_t ltm = (1ull << li) - 1ull;
            int32_t pre0 = 0, pre1 = 0;
            for (;;) {
              n_relax += 1;
              const bool win0 = act && pre0 < T0;
              const bool win1 = act && !win0 && pre1 < T1;
              const int32_t np0 = kasw::popc((kasw::ballot(win0) >> gsh) & ltm);
              const int32_t np1 = kasw::popc((kasw::ballot(win1) >> gsh) & ltm);
              const bool moved = act && (np0 != pre0 || np1 != pre1);
              pre0 = np0; pre1 = np1;
              if (kasw::ballot(moved) == 0ull) break;
            }
            const int32_t back = kasw::shfl(pre0 | (pre1 << 8), member ? gsh + (int32_t)kx : lane);   // rank view -> owner
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              f0[q] = (member && hx == q) ? bx0 + (uint32_t)(back & 0xff) : f0[q];
              f1[q] = (member && hx == q) ? bx1 + (uint32_t)((back >> 8) & 0xff) : f1[q];
            }
            n_run_rows += (member && kx > 0u) ? 1 : 0;
            ready = ready || member;
          }
        }
      }
      int32_t w0, w1;
      ticket_picks(f0, f1, meta, w0, w1);
      const int32_t w2 = 3 - w0 - w1;
      if (KAS_ORDER_DIAG && a.stats) n_rdy += kasw::popc((kasw::ballot(ready) >> (g * GL)) & (GL == 64 ? ~0ull : ((1ull << GL) - 1ull)));
      if (ready) {
        const int32_t Lp = (meta >> 3) & 3;
        const int32_t ad0 = (w0 == 0 ? e0 : (w0 == 1 ? e1 : e2)) & 0xffff;
        const int32_t ad1 = (w1 == 0 ? e0 : (w1 == 1 ? e1 : e2)) & 0xffff;
        const int32_t ad2 = (w2 == 0 ? e0 : (w2 == 1 ? e1 : e2)) & 0xffff;
        // updateCountersFromList (KAS:254-261): count[node][r] += 1, commits += 1
        // (padding holders get + 0: their row must keep commits == 0)
        if constexpr (PK) {
          kasw::lds_atomic_add((int*)(lds_raw + ad0), Lp > 0 ? 1 : 0);
          kasw::lds_atomic_add((int*)(lds_raw + ad1), Lp > 1 ? (1 << 10) : 0);
          kasw::lds_atomic_add((int*)(lds_raw + ad2), Lp > 2 ? (1 << 20) : 0);
        } else {
          kasw::lds_atomic_add_u64((uint64_t*)(lds_raw + ad0), Lp > 0 ? 1ull + (1ull << 48) : 0ull);
          kasw::lds_atomic_add_u64((uint64_t*)(lds_raw + ad1), Lp > 1 ? (1ull << 16) + (1ull << 48) : 0ull);
          kasw::lds_atomic_add_u64((uint64_t*)(lds_raw + ad2), Lp > 2 ? (1ull << 32) + (1ull << 48) : 0ull);
        }
        ring[my_slot].tag = KAS_TAG_DONE | w0 | (w1 << 2) | (Lp << 4);
        cv = false;
      }
      {
        // lanes without a row claim the next rows of their group in order; a claimed row is taken if
        // its tile has been staged (tiles are staged whole, so the rows taken are a prefix of the
        // rows claimed and gnext moves on by their number); the END tile ends the group
        constexpr uint64_t GLM = GL == 64 ? ~0ull : ((1ull << GL) - 1ull);
        const bool need = !cv && !gfin;
        const uint64_t nbg = (kasw::ballot(need) >> (g * GL)) & GLM;
        const int32_t v = gnext + kasw::popc(nbg & ((1ull << li) - 1ull));
        const int32_t tile = v / GL, slot = (tile & (K - 1)) * 64 + g * GL + (v % GL);
        // (the slot is used through selects, not behind a branch on its tag: the compiler then keeps the
        // ONE 16-byte LDS read — behind a branch it read the tag, tested it and fetched the row in a second,
        // dependent round trip in every step of the chain.  The staging wave writes a slot with one 16-byte
        // store, so a tag that says "staged" never comes with an older row.)
        const RingSlot sl = ring[need ? slot : my_slot];
        const bool taken = need && sl.tag >= 0 && (sl.tag & KAS_TAG_JMASK) == (tile & KAS_TAG_JMASK);
        const bool saw_end = need && sl.tag == KAS_TAG_END;
        e0 = taken ? sl.c[0] : e0; e1 = taken ? sl.c[1] : e1; e2 = taken ? sl.c[2] : e2;
        meta = taken ? sl.tag >> 26 : meta;
        my_slot = taken ? slot : my_slot;
        cv = cv || taken;
        gnext += kasw::popc((kasw::ballot(taken) >> (g * GL)) & GLM);
        const uint64_t endb = kasw::ballot(saw_end);         // (a collective: not behind a short-circuit)
        gfin = gfin || ((endb >> (g * GL)) & GLM) != 0ull;
      }
      const bool fin = gfin && !cv;
#if KAS_STARVE_NAP > 0
      // Lanes that wanted a row and found none staged: the staging wave is behind.  Stepping on with a
      // thin hand costs a full step's instructions for a few rows — at raised priority, i.e. taken from
      // the very waves that have to catch up — so the solver lets them.
      if (kasw::popc(kasw::ballot(!cv && !gfin)) >= KAS_STARVE_MIN) kasw::nap<KAS_STARVE_NAP>();
#endif
      if (kasw::ballot(!fin) == 0) break;
      const bool progress = kasw::ballot(ready) != 0;
      // (the solver does not count for the watchdog: a step that decided something has nothing of it on its path —
      // as a counter it cost a lone scenario 8 %, configs[1] 0.87 -> 0.945 ms — and a stuck workgroup has its
      // staging and retiring waves polling idle, which raise the word; the solver looks at it when blocked)
      if (!progress) {
        n_blocked += 1;
        if (watchdog_raised(wd, (int32_t)n_blocked)) break;
        kasw::spin_pause();
      }
    }
    int32_t run_rows = 0;                                  // rows of my scenario decided inside runs
    if (a.stats) {
#pragma unroll
      for (int gg = 0; gg < G; ++gg) {
        const int32_t v = kasw::wave_sum(g == gg ? (int32_t)n_run_rows : 0);
        run_rows = g == gg ? v : run_rows;
      }
    }
    if (a.stats && have_s && li == 0) {
      int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
      st[8] = kasw::clock_ticks() - t_begin; st[9] = n_iter; st[10] = n_relax; st[11] = n_blocked;
      st[14] = run_rows; st[6] = n_runs; st[15] = n_cur;
      if (KAS_ORDER_DIAG) st[7] = n_rdy;
    }
  } else if (wave == 1) {
    // ------------------------------------------------------------------ stager: tickets + staging
    // Every lane runs the same LDS instructions each iteration: a lane without a row, or a list
    // position without a holder, works on the padding node nmax (its mask / count words are
    // scratch), so nothing below branches on the data.
    const uint64_t gmask = (G == 1 ? ~0ull : ((1ull << GL) - 1ull)) << (g * GL);   // my group's lanes
    const uint32_t mybit = 1u << (li & 31);
    const uint32_t lt = mybit - 1u;
    const uint32_t pad = (uint32_t)nmax;
    const int32_t dummy_tk = PK ? 3 * 0x3ff : 0;
    TileIter itl = tile_iter_begin(have_s);
#ifdef KAS_STAGER_PRIO
    kasw::set_priority<KAS_STAGER_PRIO>();                  // tuning builds: the staging wave above the retiring one
#endif
    int32_t jl = 0;                                         // tiles staged (group-uniform)
    bool endl = false;
    bool pf_valid = false, pf_end = false;
    uint32_t pf_w0 = ~0u, pf_w1 = ~0u;                      // the read-ahead tile's mid row as two dwords of uint16
                                                            // cells, AS LOADED (unpacked when it is staged)
    int32_t pf_rot = KAS_ROT_IDENT;
    const uint16_t* rowp = (const uint16_t*)a.out;          // my row of the read-ahead tile (mid rows)
    int64_t f_iter = 0, f_idle = 0;
    for (;;) {
      kasw::repoll();
      f_iter += 1;
      // ---- next tile of my group: its HBM read was issued an iteration ago (pf_*); it is staged
      // now if the slot it goes to is free (retired) in every lane of the group
      const bool slot_free = ring[(jl & (K - 1)) * 64 + lane].tag == KAS_TAG_FREE;
      const bool room = (kasw::ballot(slot_free) & gmask) == gmask;
      const bool stalled = KAS_TEST_STALL_AFTER > 0 && jl >= KAS_TEST_STALL_AFTER;   // debug-build test hook
      const bool staging = !endl && room && pf_valid && !stalled;
      const bool staging_end = staging && pf_end;
      // unpack: uint16 node indices, 0xffff = none (sorts last, like ~0)
      const uint32_t w0m = staging ? pf_w0 : ~0u, w1m = staging ? pf_w1 : ~0u;
      const uint32_t c0 = w0m & 0xffffu, c1 = w0m >> 16, c2 = w1m & 0xffffu;
      const int32_t rot = pf_rot;
      endl = endl || staging_end;
      pf_valid = pf_valid && !staging;
      // read ahead: the tile after that.  The usual step — the next tile of the same topic, rows of the batch's
      // width — is straight-line code: the iterator moves on by selects, the lane's row pointer by a constant, and
      // ONE predicate guards the load; a topic change, the end of the rows and rows narrower than the batch
      // share a side branch the whole wave skips (nested per-group ifs cost this wave ~30 scalar instructions
      // of exec-mask bookkeeping per tile, and its instruction stream is the kernel's pace)
      constexpr int MWC = mid_width_of<W>();
      const bool want = !pf_valid && !endl;
      const int32_t nrow0 = itl.row0 + GL;
      const bool usual = W >= 2 && want && !itl.exhausted && nrow0 < itl.tP && itl.tow == W;
      pf_valid = pf_valid || want;
      pf_w0 = want ? ~0u : pf_w0; pf_w1 = want ? ~0u : pf_w1;
      itl.row0 = usual ? nrow0 : itl.row0;
      rowp = usual ? rowp + GL * MWC : rowp;
      if (usual && nrow0 + li < itl.tP) {
        // the fill kernel's mid row; nothing is computed from it before it is staged
        if constexpr (W == 3) {
          pf_w0 = load_u32_a2(rowp); pf_w1 = (uint32_t)rowp[2] | 0xffff0000u;
        } else if constexpr (W == 2) {
          pf_w0 = load_u32_a2(rowp);
        }
      }
      if (want && !usual) {                                 // (rare)
        if (tile_next<GL>(itl, a, sd)) {
          const int32_t p = itl.row0 + li;
          pf_rot = itl.rot;
          rowp = (const uint16_t*)a.out + tile_mid_offset(itl) + (int64_t)p * mid_width(itl.tow);
          if (p < itl.tP) {
            if (W == 3 && itl.tow == 3) {
              pf_w0 = load_u32_a2(rowp); pf_w1 = (uint32_t)rowp[2] | 0xffff0000u;
            } else if (W == 2 && itl.tow == 2) {
              pf_w0 = load_u32_a2(rowp);
            } else {                                        // narrower rows of a wider batch: cell by cell
              uint32_t v0 = (uint32_t)rowp[0], v1 = 0xffffu, v2 = 0xffffu;
              if (W > 1 && itl.tow > 1) v1 = (uint32_t)rowp[1];
              if (W > 2 && itl.tow > 2) v2 = (uint32_t)rowp[2];
              pf_w0 = v0 | (v1 << 16); pf_w1 = v2 | 0xffff0000u;
            }
          }
        } else {
          pf_end = true;
        }
      }
      // nothing to stage for either group (ring full or rows still on their way): skip the ticket work
      // of this iteration — a third of the iterations, and their instructions are issue slots taken
      // from the solvers sharing the CU — and poll again after a nap
      if (kasw::ballot(staging) == 0) {
        if (kasw::ballot(!endl) == 0) break;
        if (watchdog_poll(wd, false, wd_idle)) break;
        f_idle += 1;
        kasw::nap<KAS_IDLE_NAP>();
        continue;
      }
      // ---- holders ascending (Sets.newTreeSet, KAS:228); empty cells (-1) sort last
      const uint32_t ab_lo = c0 < c1 ? c0 : c1, ab_hi = c0 < c1 ? c1 : c0;
      const uint32_t s0 = ab_lo < c2 ? ab_lo : c2;
      const uint32_t s2 = ab_hi < c2 ? c2 : ab_hi;
      const uint32_t mid_hi = ab_hi < c2 ? ab_hi : c2;
      const uint32_t s1 = ab_lo < mid_hi ? mid_hi : ab_lo;
      // holders of the row: a cell is a node index (< 32768, a plan limit) or KAS_MID_NONE — bit 15 tells them apart
      const int32_t Lp = 3 - __builtin_popcount(w0m & 0x80008000u) - (int32_t)((w1m >> 15) & 1u);
      uint32_t hn[3] = {s0 < pad ? s0 : pad, s1 < pad ? s1 : pad, s2 < pad ? s2 : pad};
      // ---- tickets for the tile (wave-wide lockstep).  One 32-bit lane mask per node: a 64-lane
      // group takes its tile as two ascending halves.
      uint32_t tk[3] = {0u, 0u, 0u};
#pragma unroll
      for (int hf = 0; hf < HALVES; ++hf) {
        const bool mine = HALVES == 1 || (li >> 5) == hf;
        uint32_t nn[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) nn[q] = mine ? hn[q] : pad;
#pragma unroll
        for (int q = 0; q < 3; ++q) kasw::lds_atomic_or_u32(&dep[nn[q]], mybit);
        kasw::lockstep();
        uint32_t m[3], base[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { m[q] = dep[nn[q]]; base[q] = (uint32_t)run[nn[q]]; }
        kasw::lockstep();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const uint32_t t = base[q] + (uint32_t)__builtin_popcount(m[q] & lt);
          tk[q] = mine ? t : tk[q];
          // the lowest lane holding the node moves its running count on and clears the mask
          const uint32_t wn = (m[q] & lt) == 0u ? nn[q] : pad;
          run[wn] = (uint16_t)(base[q] + (uint32_t)__builtin_popcount(m[q]));
          dep[wn] = 0u;
        }
        if (hf + 1 < HALVES) kasw::lockstep();               // the second half sees the first's counts
      }
      // ---- hand the staged row to the solver: stored[t] = ticket << 16 | LDS byte address of the
      // holder's counter row, holders in the first pick's visit order (padding behind them)
      if (staging) {
        int32_t enc[3];
#pragma unroll
        for (int q = 0; q < 3; ++q)
          enc[q] = (int32_t)(((q < Lp ? tk[q] : (uint32_t)dummy_tk) << 16) | (uint32_t)(cnt_base + (int32_t)hn[q] * RB));
        // (a list of one holder may take the two-holder order as well: whichever of the first two stored positions
        // it lands in, the padding holders beside it never win a pick — one shift instead of a three-way choice
        // that the compiler turns into exec-mask branches)
        const int32_t lut = (rot >> (Lp == 3 ? 0 : 6)) & 0x3f;
        RingSlot o;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int32_t r = (lut >> (2 * t)) & 3;
          o.c[t] = r == 0 ? enc[0] : (r == 1 ? enc[1] : enc[2]);
        }
        // a row nobody holds (KAS:205-214 never lists it), or no row of this tile for my lane: an
        // empty list (Lp = 0, padding holders only) keeps the lane's row counters in step
        const int32_t bits = Lp == 3 ? ((rot >> 12) & 7) : 0;
        o.tag = staging_end ? KAS_TAG_END : ((jl & KAS_TAG_JMASK) | (bits << 26) | (Lp << 29));
        ring[(jl & (K - 1)) * 64 + lane] = o;
        jl += 1;
      }
      if (kasw::ballot(!endl) == 0) break;
      if (watchdog_poll(wd, true, wd_idle)) break;
    }
    if (a.stats && have_s && li == 0) {
      int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
      st[12] = f_iter; st[13] = f_idle;
    }
  } else {
    // ------------------------------------------------------------------ retirer: finished rows ->
    // broker ids, digest, the final out row.  A finished slot is copied to registers and freed for
    // the stager at once; the node index -> broker id reads (L2) of two batches of UR rows per
    // lane are in flight alternately, so their latency is not in the slot's way.
    constexpr int UR = KAS_RETIRE_UR;                       // rows per lane per batch
    constexpr int RSH = PK ? 2 : 3;                         // log2(bytes per counter row)
    struct Retired { bool on; int32_t id[3], Lp, p, kw; int32_t* row; };   // kw = topic | row width << 27
    TileIter itr = tile_iter_begin(have_s);
#ifdef KAS_RETIRER_PRIO
    kasw::set_priority<KAS_RETIRER_PRIO>();                 // tuning builds
#endif
    int32_t jr = 0;
    bool fin = false;
    uint64_t digest = 0;
    auto finish = [&](Retired& r) {
      if (r.on) {
#pragma unroll
        for (int q = 0; q < W; ++q)
          if (q < r.Lp) digest += kas_digest_cell((uint32_t)r.kw & 0x7ffffffu, (uint32_t)r.p, (uint32_t)q, r.id[q]);
#ifndef KAS_TUNE_NO_ROW_STORES
        if (r.Lp == W && (r.kw >> 27) == W) {                 // the usual row: full width, one W-dword store
          RowW<W> o;
#pragma unroll
          for (int q = 0; q < W; ++q) o.v[q] = r.id[q];
          *reinterpret_cast<RowW<W>*>(r.row) = o;
        } else {
#pragma unroll
          for (int q = 0; q < W; ++q) if (q < r.Lp) r.row[q] = r.id[q];
          // -1 behind a list shorter than the row (rare: the mid row this came from lives elsewhere)
#pragma unroll
          for (int q = 0; q < W; ++q) if (q >= r.Lp && q < (r.kw >> 27)) r.row[q] = -1;
        }
#endif
      }
      r.on = false;
    };
    auto gather = [&](Retired& r) -> bool {
      if (fin) return false;
      const RingSlot sl = ring[(jr & (K - 1)) * 64 + lane];
      if (KAS_TAG_IS_DONE(sl.tag)) {
        ring[(jr & (K - 1)) * 64 + lane].tag = KAS_TAG_FREE;
        jr += 1;
        tile_next<GL>(itr, a, sd);
        const int32_t w0 = sl.tag & 3, w1 = (sl.tag >> 2) & 3;
        const int32_t w[3] = {w0, w1, 3 - w0 - w1};
        r.on = true; r.Lp = (sl.tag >> 4) & 3; r.p = itr.row0 + li;
        r.kw = itr.k | ((r.p < itr.tP ? itr.tow : 0) << 27);   // a lane past the last row of the tile writes nothing
        r.row = a.out + itr.tout + (int64_t)r.p * itr.tow;
#pragma unroll
        for (int q = 0; q < W; ++q) {
          const int32_t e = w[q] == 0 ? sl.c[0] : (w[q] == 1 ? sl.c[1] : sl.c[2]);
          const int32_t node = q < r.Lp ? ((e & 0xffff) - cnt_base) >> RSH : 0;
          r.id[q] = g_node_id[node];                      // 4 KB table per scenario: L2-resident
        }
        return true;
      }
      if (sl.tag == KAS_TAG_END) fin = true;
      return false;
    };
    Retired ra[UR], rb[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      ra[u].on = false; rb[u].on = false;
      ra[u].Lp = 0; rb[u].Lp = 0; ra[u].p = 0; rb[u].p = 0; ra[u].kw = 0; rb[u].kw = 0;
      ra[u].row = nullptr; rb[u].row = nullptr;
#pragma unroll
      for (int q = 0; q < 3; ++q) { ra[u].id[q] = 0; rb[u].id[q] = 0; }
    }
    int64_t r_iter = 0, r_idle = 0;
    for (;;) {
      bool retired = false;
      kasw::repoll();
      r_iter += 1;
#pragma unroll
      for (int u = 0; u < UR; ++u) finish(ra[u]);
#pragma unroll
      for (int u = 0; u < UR; ++u) retired = gather(ra[u]) || retired;
      kasw::repoll();
#pragma unroll
      for (int u = 0; u < UR; ++u) finish(rb[u]);
#pragma unroll
      for (int u = 0; u < UR; ++u) retired = gather(rb[u]) || retired;
      if (kasw::ballot(!fin) == 0) break;
      const bool progress = kasw::ballot(retired) != 0;
      if (watchdog_poll(wd, progress, wd_idle)) break;
      if (!progress) { r_idle += 1; kasw::nap<KAS_IDLE_NAP>(); }
    }
    if (KAS_ORDER_DIAG && a.stats && have_s && li == 0) {
      int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
      st[4] = r_iter; st[5] = r_idle;
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) { finish(ra[u]); finish(rb[u]); }
    kasw::lds_atomic_add_u64(&gdig[g], digest);
    kasw::lockstep();
    if (have_s && li == 0) a.scenario_results[s].digest = gdig[g];
    // the Context goes back (KAS:360-369): every row of the scenario has been retired, so every commit
    // is in the counter rows (the solver adds before it marks a slot done)
    if constexpr (!PK) {
      for (int32_t n = li; n < N && ccols > 0; n += GL) {
        const uint64_t x = ((const uint64_t*)cnt)[n];
        for (int32_t r = 0; r < ccols; ++r) g_ctx[(int64_t)n * sd.ctx_width + r] = (int32_t)((x >> (16 * r)) & 0xffffu);
      }
    }
    if (KAS_SPIN_BOUND > 0 && have_s && li == 0 && *(volatile uint32_t*)wd != 0u) {
      a.scenario_results[s].status = KAS_FAIL_WATCHDOG;
      a.scenario_results[s].fail_topic = -1; a.scenario_results[s].fail_partition = -1;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// order kernel, round form (one wavefront per scenario; int32 counters, Context in/out).
// 64 ascending rows per tile; once per tile every lane ors its lane bit into a 64-bit LDS mask
// per node it holds and reads those masks back, so it knows which LOWER lanes share a node with
// it; each round a lane none of whose lower sharers is still pending commits.  Lane order ==
// row order, so the result is the sequential one.  Returns true on the KAS:190 index error.
// ---------------------------------------------------------------------------------------------
template <int W>
KAS_DEV bool order_rounds(int32_t* cnt, uint64_t* dep, const kas_topic_desc& td, const OutRef& out, const uint16_t* mid,
                          const int32_t* g_node_id, uint32_t topic_k, uint64_t& digest,
                          int64_t& rounds) {
  constexpr int CS = cnt_stride<W>();
  const int lane = kasw::lane();
  const uint64_t lt = kasw::lanemask_lt();
  const int32_t P = td.n_partitions, ow = td.out_width, nt = (P + 63) >> 6;
  const int32_t hash = td.name_hash;
  const bool hash_min = hash == (int32_t)0x80000000;
  const uint64_t mybit = 1ull << lane;
  // rotation offsets idx_m = Math.abs(hash) % m for every set size m (KAS:190, via KAS:267)
  int32_t idxm[W + 1];
#pragma unroll
  for (int m = 1; m <= W; ++m) idxm[m] = java_abs_mod(hash, m);
  idxm[0] = 0;
  // rows come in as mid rows at the end of the topic's out region and leave as final rows from its
  // start: the tile after the current one is read before the current tile's final rows are written
  // (16-bit cells: final row p takes mid row p's place — read a tile ago)
  MidRaw<W> nxr = mid_load_raw<W>(mid, ow, lane < P ? lane : 0, lane < P);   // next tile's mid row (software prefetch)
  for (int32_t tile = 0; tile < nt; ++tile) {
    const int32_t p = (tile << 6) + lane;
    const bool active = p < P;
    int32_t nx[W], h[W], Lp;
    mid_unpack<W>(nxr, ow, nx);
    sort_holders<W>(nx, h, Lp);
    {
      const int32_t pn = p + 64;
      nxr = mid_load_raw<W>(mid, ow, pn < P ? pn : 0, pn < P);
    }
    if (hash_min) {
      // KAS:190 index error: some set size m <= L has a negative rotation offset
      bool bad = false;
#pragma unroll
      for (int m = 1; m <= W; ++m) bad = bad || (m <= Lp && idxm[m] < 0);
      if (kasw::ballot(bad) != 0) return true;
    }
    bool pending = active && Lp > 0;
    // node index per list position, clamped so that unused positions address node 0
    int32_t hn[W];
#pragma unroll
    for (int k = 0; k < W; ++k) hn[k] = (pending && k < Lp) ? h[k] : 0;
    if (pending) {
#pragma unroll
      for (int k = 0; k < W; ++k) if (k < Lp) kasw::lds_atomic_or_u64(&dep[hn[k]], mybit);
    }
    kasw::lockstep();
    uint64_t share = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) share |= (pending && k < Lp) ? dep[hn[k]] : 0ull;
    kasw::lockstep();
    if (pending) {
#pragma unroll
      for (int k = 0; k < W; ++k) if (k < Lp) dep[hn[k]] = 0ull;
    }
    const uint64_t depmask = share & lt;

    int32_t lst[W];
#pragma unroll
    for (int k = 0; k < W; ++k) lst[k] = -1;
    for (;;) {
      const uint64_t pend = kasw::ballot(pending);
      if (pend == 0) break;
      rounds += 1;
      const bool ready = pending && (depmask & pend) == 0;
      // count[node][0..W) of my nodes (KAS:280-301); every lane computes, ready lanes commit
      int32_t c[W][W];
#pragma unroll
      for (int k = 0; k < W; ++k) load_cnt_row<W>(c[k], cnt + hn[k] * CS);
      int32_t pos[W], cnt_r[W];
      pick_row<W>(c, Lp, pending, idxm, pos, cnt_r);
      if (ready) {
#pragma unroll
        for (int r = 0; r < W; ++r) {
          if (r < Lp) {
            const int32_t node = sel<W>(hn, pos[r]);
            lst[r] = node;
            cnt[node * CS + r] = cnt_r[r] + 1;             // updateCountersFromList (KAS:254-261)
          }
        }
        pending = false;
      }
      kasw::lockstep();
    }
    if (active) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        if (k < ow) {
          const int32_t node = lst[k];
          const int32_t id = node >= 0 ? (out.o16 ? node : g_node_id[node]) : -1;
          out_store(out, (int64_t)p * ow + k, id);
          if (node >= 0) digest += kas_digest_cell(topic_k, (uint32_t)p, (uint32_t)k, id);
        }
      }
    }
  }
  return false;
}

template <int W>
KAS_DEV void order_scenario_rounds(const KasLaunch& a, int32_t s, unsigned char* lds_raw) {
  constexpr int CS = cnt_stride<W>();
  const int lane = kasw::lane();
  if ((a.flags & KAS_FLAG_ORDER_FLAGGED) && a.ord_flag[s] == 0) return;   // a ticket form did this one
  const kas_scenario_desc sd = a.scen[s];
  const int32_t N = sd.n_nodes;
  int32_t* cnt = (int32_t*)lds_raw;
  uint64_t* dep = (uint64_t*)(lds_raw + kas_align16(4 * (int64_t)(a.n_max > 0 ? a.n_max : 1) * CS));
  const int32_t* g_node_id = a.node_id + sd.node_off;
  const bool has_ctx = sd.ctx_off >= 0 && sd.ctx_width > 0;
  int32_t* g_ctx = has_ctx ? a.ctx + sd.ctx_off : nullptr;
  const int32_t ctxw = sd.ctx_width;
  const int64_t t_begin = kasw::clock_ticks();
  // Context counters (KAS:360-369) into LDS
  for (int32_t i = lane; i < N; i += 64) {
#pragma unroll
    for (int r = 0; r < CS; ++r)
      cnt[i * CS + r] = (has_ctx && r < W && r < ctxw) ? g_ctx[(int64_t)i * ctxw + r] : 0;
    dep[i] = 0ull;
  }
  kasw::lockstep();
  kas_scenario_result sr = a.scenario_results[s];          // written by the fill kernel
  uint64_t digest = 0;
  int64_t rounds = 0;
  bool failed_here = false;
  for (int32_t k = 0; k < sd.topic_count; ++k) {
    const int32_t ti = sd.topic_begin + k;
    kas_topic_result tr = a.topic_results[ti];
    const kas_topic_desc td = a.topics[ti];
    const OutRef out = topic_out(a, td);
    kasw::lockstep();                                      // every lane has read tr before lane 0 rewrites it
    if (failed_here && tr.status != KAS_SKIPPED) {
      // an earlier topic failed in this kernel: the CLI run would have aborted (KAG:173-184)
      if (tr.status == KAS_OK) {
        sr.moved_replicas -= tr.moved_replicas; sr.moved_partitions -= tr.moved_partitions;
        out_pad(out, (int64_t)td.n_partitions * td.out_width, lane, 64);
      }
      tr.status = KAS_SKIPPED; tr.fail_partition = -1; tr.moved_replicas = 0; tr.moved_partitions = 0;
      if (lane == 0) a.topic_results[ti] = tr;
      continue;
    }
    if (tr.status != KAS_OK) continue;
    uint64_t dg = 0;
    const bool hash_fail = order_rounds<W>(cnt, dep, td, out, topic_mid(a, td), g_node_id, (uint32_t)k, dg, rounds);
    if (hash_fail) {
      kasw::wave_sync();
      out_pad(out, (int64_t)td.n_partitions * td.out_width, lane, 64);
      sr.moved_replicas -= tr.moved_replicas; sr.moved_partitions -= tr.moved_partitions;
      tr.status = KAS_FAIL_HASH_INDEX; tr.fail_partition = -1; tr.moved_replicas = 0; tr.moved_partitions = 0;
      if (lane == 0) a.topic_results[ti] = tr;
      sr.status = KAS_FAIL_HASH_INDEX; sr.fail_topic = k; sr.fail_partition = -1;
      failed_here = true;
      for (int32_t i = lane; i < N; i += 64) dep[i] = 0ull;
      kasw::lockstep();
    } else {
      digest += dg;
    }
  }
  if (has_ctx) {
    for (int32_t i = lane; i < N; i += 64)
#pragma unroll
      for (int r = 0; r < W; ++r)
        if (r < ctxw) g_ctx[(int64_t)i * ctxw + r] = cnt[i * CS + r];
  }
  const uint64_t dsum = kasw::wave_sum_u64(digest);
  if (lane == 0) {
    sr.digest = dsum;
    a.scenario_results[s] = sr;
    if (a.stats) {
      a.stats[(int64_t)s * KAS_STATS_PER_SCENARIO + 8] = kasw::clock_ticks() - t_begin;
      a.stats[(int64_t)s * KAS_STATS_PER_SCENARIO + 6] = rounds;
    }
  }
}

}  // namespace kas

#include "kas_order_wide.h"
#include "kas_order_relax.h"
#include "kas_order_relax_wide.h"
