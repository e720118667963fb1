// kas_plan_math.h — host-side shape validation, LDS carve-ups and launch arguments.
//
// Pure C++ (no HIP calls) so that the exact same planning code runs in the product library
// (kas_hip.hip) and in the CPU emulation harness under tests/emu/.
//
// A solve is two kernels on one stream (+ a one-workgroup kernel that orders the scenarios for the
// ticket form, and — for batches of few large single-topic scenarios — the four spread-fill kernels in
// front of the fill kernel, which then only takes what they hand back):
//   fill   (P0-P4 + tickets)  one workgroup of NW wavefronts per scenario; LDS = node state
//   order  (P5)               ticket form: one lane group per scenario, G groups per wavefront,
//                             LDS = packed 16-bit counters; round form: one wavefront per
//                             scenario, LDS = int32 counters + lane masks
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "kas_abi.h"

// Kernel arguments: every pointer is a device pointer (host pointer in the emulator).
struct KasLaunch {
  const kas_scenario_desc* scen;
  const kas_topic_desc* topics;
  const int32_t* node_id;
  const int32_t* node_rack;
  const int32_t* cur;
  int32_t* out;
  const int32_t* aux;
  int32_t* ctx;
  kas_topic_result* topic_results;
  kas_scenario_result* scenario_results;
  uint64_t* accmask;            // scratch: accept-mask ballot words, one region per scenario
  const int64_t* accmask_off;   // [n_scenarios] offset of the region in 64-bit words
  int32_t* orph;                // scratch: orphan row lists, one region per scenario
  const int64_t* orph_off;      // [n_scenarios] offset of the region in int32 elements
  int32_t* perm;                // scratch: order in which the order kernel takes the scenarios, or NULL
  int32_t* ord_flag;            // [n_scenarios] != 0: a ticket-form order kernel left the scenario to the round form
                                // (a Context counter too large for its count fields), or NULL
  int64_t* stats;               // [n_scenarios][KAS_STATS_PER_SCENARIO] device counters, or NULL
  // spread fill (large single-topic scenarios: passes A and B over many one-wavefront workgroups)
  int32_t* sp_hist;             // [S][chunks][W][n_max] sweep histograms of the chunks (pass A)
  int32_t* sp_quota;            // [S][chunks][n_max]    quota of sweep r* left when the chunk starts
  int32_t* sp_node;             // [S][2][n_max]         load after the sticky fill, r* << 28 | quota
  int32_t* sp_flag;             // [S]                   != 0: the scenario takes the one-workgroup kernel instead
  int32_t* sp_oc;               // [S][chunks + 2]       orphans per chunk, then moved replicas / partitions
  int32_t sp_chunks;            // chunks per scenario (0: no spread fill in this launch)
  // split first fit (KAS_FLAG_SPLIT_P4, round 5): the fill kernel ends a rack-diverse topic behind pass B and leaves P4 to
  // kas_p4_kernel — four wavefronts on 9 KB of LDS instead of the fill workgroup's 128 VGPRs x 4 and 35 KB for the
  // 0.3 ms in which first fit touches no table
  int32_t* p4s;                 // [n_topics][KAS_P4S_HEAD + n_max]: head words (kas_plan_math.h), then the brokers' loads after P2
  int32_t n_scenarios;
  int32_t n_max;                // largest broker count in the batch (LDS array extent)
  int32_t idmap_entries;        // entries of the direct broker-id -> node-index table
  int32_t need_bsearch;         // some scenario's id range exceeds idmap_entries
  uint32_t flags;               // KAS_FLAG_*
  int32_t* handback;            // (may be null) where a fill kernel launched for flagged scenarios only leaves their number: host memory the
                                // device can write (the plan sizes its next such launch by it, kas_hip.hip: kas_plan_back_grid)
};

#define KAS_FLAG_GENERIC_FILL 1u   // always use the general sticky fill (testing / comparison)
#define KAS_FLAG_ROUND_ORDER  2u   // always use the tile-round preference ordering (testing / comparison)
#define KAS_FLAG_WIDE_COUNTERS 4u  // ticket form: always 4 x uint16 counter rows (testing / comparison)
#define KAS_FLAG_TWO_PASS_HIST 8u  // rack-diverse fill: keep the separate chunk-count pass (testing / comparison)
#define KAS_FLAG_FUSED_HIST   16u  // set by the launcher: per-chunk histograms, no chunk-count pass (KasShape::fused_ok)
#define KAS_FLAG_SPREAD_FILL  32u  // spread fill also for small scenarios, with few chunks (testing / comparison)
#define KAS_FLAG_ONLY_FLAGGED 64u  // set by the launcher: the fill kernel takes only the scenarios the spread fill handed back
#define KAS_FLAG_ORDER_FLAGGED 128u // set by the launcher: the round-form order kernel takes only scenarios with ord_flag set
#define KAS_FLAG_CELLS16      512u  // set by the launcher (plans of kas_plan_create16, ABI v5): cur / out cells are uint16 node indices (node i has id i),
                                   // out rows take the place of the mid rows they are made from
#define KAS_FLAG_INDEX_ROWS   0x400u // set by the launcher (int32 cells, per-chunk histograms, LDS lane order; not with KAS_PLAN_NO_INDEX_ROWS): the fill's
                                   // first row scan leaves every row's node indices where its mid row goes and the second scan streams those
                                   // (kas_solver_body.h, fill_pass_a_fused<EMIT>): `cur` is read once, 6 instead of 12 bytes a row the second time
#define KAS_PLAN_FULL_FILL_BIT 16u     // the user's switch (KAS_PLAN_FULL_FILL; KAS_FLAG_FUSED_HIST's bit in a launch word, so kept beside the plan's flags):
                                      // kas_fill_kernel for every scenario, no kas_fill_slim_kernel in front of it (testing / comparison)
#ifndef KAS_SLIM_FILL_DEFAULT
#define KAS_SLIM_FILL_DEFAULT 1
#endif
#define KAS_PLAN_NO_INDEX_ROWS_BIT 64u // the user's switches (kas_plan_set_flags; their bits are KAS_FLAG_ONLY_FLAGGED's / KAS_FLAG_ORDER_FLAGGED's in a launch
#define KAS_PLAN_INDEX_ROWS_BIT 128u   // word, so they are kept beside the plan's flags): off / on whatever KAS_INDEX_ROWS_DEFAULT says
#ifndef KAS_INDEX_ROWS_DEFAULT
#define KAS_INDEX_ROWS_DEFAULT 0   // (round 6, measured at the headline shape with twelve batches in flight: 643k scenarios/s with, 661k without — the stores of pass A cost more than the second read of cur)
#endif
// index rows for a plan / emulator flag word that names neither switch?
KAS_ABI_FN int32_t kas_index_rows_wanted(uint32_t user_flags) {
  if (user_flags & KAS_PLAN_NO_INDEX_ROWS_BIT) return 0;
  if (user_flags & KAS_PLAN_INDEX_ROWS_BIT) return 1;
  return KAS_INDEX_ROWS_DEFAULT;
}
#define KAS_FLAG_MID32        0x800u // set by the launcher (kas_mid32_ok; not with KAS_PLAN_NO_MID32): every mid row of this launch is ONE dword — the row's
                                   // holders SORTED by node index, 11 bits each (kas_solver_body.h, mid32_pack) — instead of three uint16 in acceptance
                                   // order: 4 instead of 6 bytes a row for the fill to write and the order kernel to read, one aligned store / load
                                   // where the packed 6-byte rows take a dword and a halfword at 2-byte alignment.  Lists exactly 3 wide at most
                                   // (Wc == 3), at most KAS_MID32_N_MAX brokers, int32 cells, the relaxation form without a Context / sampled
                                   // verification in the instances with the broker ids in the LDS.
#define KAS_MID32_N_MAX 2047         // node indices 0 .. 2046 in 11 bits, 0x7ff = no holder
#define KAS_PLAN_MID32_BIT 0x80000u     // the user's switches (kas_plan_set_flags; their bits are KAS_FLAG_RELAX_DUAL's / KAS_FLAG_LANE_ORDER's in a launch
#define KAS_PLAN_NO_MID32_BIT 0x100000u // word — both set by the launcher — so they are kept beside the plan's flags): on / off whatever KAS_MID32_DEFAULT says
#ifndef KAS_MID32_DEFAULT
#define KAS_MID32_DEFAULT 1
#endif
KAS_ABI_FN int32_t kas_mid32_wanted(uint32_t user_flags) {
  if (user_flags & KAS_PLAN_NO_MID32_BIT) return 0;
  if (user_flags & KAS_PLAN_MID32_BIT) return 1;
  return KAS_MID32_DEFAULT;
}
#define KAS_FLAG_WIDE_CHECK   256u  // set by the launcher (KasShape::wide_checked): the wide ticket form checks its count fields at the end
#define KAS_FLAG_TICKET_ORDER 0x10000u // lists <= 3 wide: the ticket form of P5 where the relaxation form would run (testing / comparison);
                                       // KAS_PLAN_GROUPS(n) and KAS_PLAN_WIDE_COUNTERS, which only mean something to the ticket form, imply it
#define KAS_FLAG_RELAX_TILES_64  0x20000u // relaxation form: tiles of 64 rows whatever the batch size (KAS_PLAN_RELAX_TILES(1))
#define KAS_FLAG_RELAX_TILES_128 0x40000u // relaxation form: double tiles whatever the batch size (KAS_PLAN_RELAX_TILES(2))
#define KAS_FLAG_RELAX_DUAL      0x80000u // set by the launcher (kas_relax_double_tiles): double tiles in this launch
#define KAS_FLAG_LANE_ORDER      0x100000u // set by the launcher: the LDS hands the lanes of one atomic instruction out in lane order here (self-test)
#define KAS_FLAG_SPLIT_P4        0x400000u // KAS_PLAN_SPLIT_P4 / set by the launcher (kas_split_p4): first fit of rack-diverse topics runs in kas_p4_kernel
#define KAS_FLAG_FILL_WITH_P4    0x800000u // KAS_PLAN_FILL_WITH_P4: first fit inside the fill workgroup whatever the batch size
#define KAS_SPLIT_P4_FROM 512            // batches of at least this many scenarios take kas_p4_kernel unless told otherwise
#define KAS_FLAG_NO_RTN_QUOTA    0x200000u // KAS_PLAN_NO_RTN_QUOTA: the fill draws its quota without the atomic-with-return (testing / comparison)
#define KAS_RELAX_DUAL_BELOW 512         // batches of fewer scenarios than this take double tiles unless told otherwise

// Byte offsets into the dynamic LDS of the fill kernel.
//   x       sweep histogram hist[W][n], then per-chunk quota qc[NW][n]
//   load, qrs, rack, idmap, ids, ring: node state of P2-P4; live (P4) overlays qrs (P2)
//   ctl     control words
struct KasLds {
  int32_t off_x;       // int32  [max(W,NW)][n_max]
  int32_t off_load;    // int32  [n_max]        |Node.assignedPartitions|
  int32_t off_qrs;     // int32  [n_max]        saturating sweep r* << 28 | quota in that sweep
  int32_t off_rack;    // int16  [n_max]        dense rack index per node
  int32_t off_live;    // int16  [n_max]        non-full nodes in processing order (== off_qrs)
  int32_t off_idmap;   // int16  [idmap_entries] broker id - min_id -> node index
  int32_t off_ids;     // int32  [n_max]        sorted ids for binary search (only if needed)
  int32_t off_ring;    // orphan window: p[128] int32, meta[128] int32, rack[W][128] int16
  int32_t off_ctl;     // int32  [KAS_CTL_INTS]
  int32_t total;
};

#define KAS_RING_CAP 128
#define KAS_LDS_LIMIT (160 * 1024)
#define KAS_IDMAP_CAP 16384
#define KAS_N_LIMIT 32767
#define KAS_MAX_WAVES 8
#define KAS_CTL_INTS 32
// control words
#define KAS_CTL_VIOL 0        // some row's replicas are not rack-diverse / node table invalid
#define KAS_CTL_FAILROW 1     // first row P4 could not place (KAS:183-184), or -1
#define KAS_CTL_FAILWIN 2     // P4 window that row was in (INT32_MAX: none)
#define KAS_CTL_LIVE 6        // P4: length of the live list
#define KAS_CTL_HEAD 5        // P4: live-list positions before this one are full
#define KAS_CTL_PROG 16       // P4: uint64 [NW] per wave (window << 32 | live-list positions done)
#define KAS_CTL_MOVED_R 3
#define KAS_CTL_MOVED_P 4
#define KAS_CTL_OC 8          // [NW] orphans found per chunk
#define KAS_CTL_WATCHDOG 7    // a P4 wait ran past its bound (KAS_SPIN_BOUND)

// wavefronts of the spread fill's P4 kernel (one workgroup per scenario).  A window of 64 orphans costs its wave
// ~3.5 us (row reads, rack lookups, two hand-over steps), the hand-over chain 0.44 us per step: with four waves
// the waves were the pace (configs[4]: 3.4k windows, 3.0 ms), not the chain; KAS_CTL_OC / KAS_CTL_PROG hold 8.
#ifndef KAS_SPREAD_P4_WAVES
#define KAS_SPREAD_P4_WAVES 8
#endif

#define KAS_TICKET_LIMIT 65535  // tickets (= 16-bit counters of the order kernel) stay below this

KAS_ABI_FN int32_t kas_align16(int64_t v) { return (int32_t)((v + 15) & ~(int64_t)15); }

// int32 counter row stride of the round form: 3-wide rows are padded to one 16-byte LDS read
KAS_ABI_FN int32_t kas_cnt_stride(int32_t W) { return W == 3 ? 4 : W; }

// Fused histogram layout of the fill kernel (with_x = 2): ONE block of kas_fused_block_words() dwords per
// node holds everything the kernel keeps per node:
//   pass A      words 0 .. ceil(NW W / 2) - 1: uint16 hist[NW][W], candidates per chunk and sweep
//   after Q     words 0 .. NW - 1: int32 quota left when chunk w starts; word NW: load; word NW + 1: qrs
//   always      low half of the last word: the node's rack index (int16)
// The block length is odd, so the words of consecutive nodes fall on different LDS banks (a stride of 6
// dwords reaches only every other bank: twice the conflicts on every node-indexed access).
KAS_ABI_FN int32_t kas_fused_block_words(int32_t W, int32_t NW) {
  const int32_t h = (NW * W + 1) / 2;
  const int32_t b = (h > NW + 2 ? h : NW + 2) + 1;
  return b | 1;
}

// with_x = 0: no histogram / quota table (only the general sticky fill is possible then)
//          1: hist[W][n] int32, then qc[NW][n] int32 over the same words (two passes over cur + a
//             chunk-count pass); load[], qrs[], rack[] are arrays of their own
//          2: fused — node-major blocks (above): uint16 hist[n][NW][W] counted per chunk in the first
//             pass, then int32 qc[n][NW] (no chunk-count pass), load, qrs and rack inside the block
KAS_ABI_FN KasLds kas_fill_lds_layout(int32_t n_max, int32_t W, int32_t NW, int32_t idmap_entries,
                                         int32_t need_bsearch, int32_t with_x) {
  KasLds L;
  int64_t n = n_max > 0 ? n_max : 1;
  int64_t o = 0;
  const int64_t xr = with_x == 2 ? kas_fused_block_words(W, NW) : (with_x ? (W > NW ? W : NW) : 0);
  L.off_x = (int32_t)o;     o = kas_align16(o + 4 * n * xr);
  if (with_x == 2) {
    // node state inside the blocks; P4's list of non-full nodes gets an array of its own
    L.off_load = L.off_x + 4 * NW;
    L.off_qrs = L.off_x + 4 * (NW + 1);
    L.off_rack = L.off_x + 4 * (int32_t)(xr - 1);
    L.off_live = (int32_t)o;  o = kas_align16(o + 2 * n);
  } else {
    // lists wider than the workgroup has waves leave histogram rows NW..W-1 unused once the quota
    // pass has consumed them: load[] (written by that pass, per node, after its reads) lives there
    if (with_x == 1 && W > NW) L.off_load = L.off_x + (int32_t)(4 * n * NW);
    else { L.off_load = (int32_t)o;  o = kas_align16(o + 4 * n); }
    L.off_qrs = (int32_t)o;   o = kas_align16(o + 4 * n);
    L.off_rack = (int32_t)o;  o = kas_align16(o + 2 * n);
    L.off_live = L.off_qrs;                      // P4's list of non-full nodes: qrs is dead after pass B
  }
  L.off_idmap = (int32_t)o; o = kas_align16(o + 2 * (int64_t)(idmap_entries > 0 ? idmap_entries : 1));
  L.off_ids = (int32_t)o;   if (need_bsearch) o = kas_align16(o + 4 * n);
  L.off_ring = (int32_t)o;  o = kas_align16(o + KAS_RING_CAP * (4 + 4 + 2 * (int64_t)W));
  L.off_ctl = (int32_t)o;   o = kas_align16(o + 4 * KAS_CTL_INTS);
  L.total = (int32_t)o;
  return L;
}

// The slim fill kernel's layout (kas_fill_slim_kernel: per-chunk histograms on 4 wavefronts, a direct id table, first fit handed
// over): the node blocks, the id table and the control words — no list of non-full nodes, no orphan ring, no sorted ids.  At the
// headline's 1,050 brokers that is 31.6 KB where the full layout has 35.5: FIVE workgroups fit a CU's 160 KB instead of four.
KAS_ABI_FN KasLds kas_fill_slim_lds(int32_t n_max, int32_t W, int32_t idmap_entries) {
  KasLds L;
  int64_t n = n_max > 0 ? n_max : 1;
  int64_t o = 0;
  const int64_t xr = kas_fused_block_words(W, 4);
  L.off_x = (int32_t)o;     o = kas_align16(o + 4 * n * xr);
  L.off_load = L.off_x + 4 * 4;
  L.off_qrs = L.off_x + 4 * (4 + 1);
  L.off_rack = L.off_x + 4 * (int32_t)(xr - 1);
  L.off_idmap = (int32_t)o; o = kas_align16(o + 2 * (int64_t)(idmap_entries > 0 ? idmap_entries : 1));
  L.off_ctl = (int32_t)o;   o = kas_align16(o + 4 * KAS_CTL_INTS);
  L.off_live = L.off_ctl; L.off_ids = L.off_ctl; L.off_ring = L.off_ctl;   // (not part of this layout: never touched)
  L.total = (int32_t)o;
  return L;
}

// Layouts of the spread fill's scan kernels (one wavefront per workgroup, many workgroups per scenario): only
// what the pass touches, so that two of them fit a CU at 5,000 brokers (the full layout above is 145 KB there
// and ran the scans of a 64-scenario batch at one wavefront per CU).
//   pass A (mode 1)  x = uint16 hist[W][n] (a chunk has fewer than 65,536 rows: kas_spread_chunks), rack, idmap
//   pass B (mode 2)  x = int32 quota[n] of the chunk, qrs[n], rack, idmap
//   P4     (mode 3)  load[n] (in x), the list of non-full nodes (in qrs), rack — no id tables
KAS_ABI_FN KasLds kas_spread_scan_lds(int32_t n_max, int32_t W, int32_t idmap_entries, int32_t need_bsearch, int32_t mode) {
  KasLds L;
  int64_t n = n_max > 0 ? n_max : 1;
  int64_t o = 0;
  L.off_x = (int32_t)o;     o = kas_align16(o + (mode == 1 ? 2 * n * W : 4 * n));
  L.off_load = L.off_x;                                    // (P4: the loads; not used by the scans)
  L.off_qrs = (int32_t)o;   if (mode == 2) o = kas_align16(o + 4 * n);
  if (mode == 3) o = kas_align16(o + 2 * n);               // (P4: int16 live[n])
  L.off_rack = (int32_t)o;  o = kas_align16(o + 2 * n);
  L.off_live = L.off_qrs;
  if (mode == 3) { idmap_entries = 0; need_bsearch = 0; }
  L.off_idmap = (int32_t)o; o = kas_align16(o + 2 * (int64_t)(idmap_entries > 0 ? idmap_entries : 1));
  L.off_ids = (int32_t)o;   if (need_bsearch) o = kas_align16(o + 4 * n);
  L.off_ring = (int32_t)o;                                 // (no orphan window in the scans)
  L.off_ctl = (int32_t)o;   o = kas_align16(o + 4 * KAS_CTL_INTS);
  L.total = (int32_t)o;
  return L;
}

// ticket form of order: per lane group (= scenario) one counter row per node + the padding
// holder's row (uint64 = 4 x uint16: count[node][0..2] + commits; or, packed, uint32 = three 10-bit
// counts when no node ever holds 1023 rows of the scenario), uint32 lane mask per node [n_max],
// uint16 tickets handed out per node [n_max]; a ring of KAS_RING_SLOTS 16-byte row slots per
// lane; one digest slot per group
#ifndef KAS_RING_SLOTS
#define KAS_RING_SLOTS 4
#endif
#define KAS_PACKED_TICKET_LIMIT 1023
// relaxation form (kas_order_relax.h): 16-bit count fields, no tickets — a node holds fewer rows than this
#define KAS_RELAX_ROW_LIMIT 65535
// wide ticket form (kas_order_wide.h): the commits of a node are an 11-bit field, and count field 4 — 11 bits,
// next to it — must not carry into them whatever the counts do: a node holds fewer rows than this.  Between
// KAS_PACKED_TICKET_LIMIT and this bound the 10-bit count fields are not safe a priori; the kernel checks them
// when the last row has retired (KasShape::wide_checked).
#define KAS_WIDE_COMMIT_LIMIT 2040
KAS_ABI_FN int32_t kas_order_ticket_group_bytes(int32_t n_max, int32_t G, int32_t packed) {
  int64_t n = n_max > 0 ? n_max : 1;
  (void)G;
  return kas_align16(kas_align16((packed ? 4 : 8) * (n + 1)) + 4 * (n + 1) + 2 * (n + 1));
}
KAS_ABI_FN int32_t kas_order_ticket_lds(int32_t n_max, int32_t G, int32_t packed) {
  return kas_align16((int64_t)G * kas_order_ticket_group_bytes(n_max, G, packed) + KAS_RING_SLOTS * 64 * 16 + 8 * (int64_t)G + 256 + 16);
}
// ticket form for lists 4 and 5 wide (kas_order_wide.h), one scenario per workgroup: uint64 counter
// row per node + the padding holder's (five 10-bit counts), uint64 lane mask per node, uint16
// tickets handed out per node, a ring of KAS_RING_SLOTS 32-byte row slots per lane, digest + queue scratch
#ifndef KAS_WIDE_RING_SLOTS
#define KAS_WIDE_RING_SLOTS 8
#endif
// solver wavefronts for the rows that do not sit on a broker being filled (kas_order_wide.h)
#ifndef KAS_WIDE_BULK_SOLVERS
#define KAS_WIDE_BULK_SOLVERS 2
#endif
// nodes whose queues one solver step of the wide kernel decides together (rank -> lane scratch per node)
#ifndef KAS_WIDE_HOT
#define KAS_WIDE_HOT 2
#endif
// counter rows + lane masks + running tickets per node, the ring, the two claim lists, digest and list
// lengths, the solvers' queue scratch, the watchdog word
KAS_ABI_FN int32_t kas_order_wide_lds_core(int32_t n_max) {
  int64_t n = n_max > 0 ? n_max : 1;
  return kas_align16(2 * (int64_t)kas_align16(8 * (n + 1)) + kas_align16(2 * (n + 1)) +
                     KAS_WIDE_RING_SLOTS * 64 * 32 + 2 * KAS_WIDE_RING_SLOTS * 64 * 2 + 16 + 256 * KAS_WIDE_HOT * (1 + KAS_WIDE_BULK_SOLVERS) + 16);
}
// front[] of the class-1 solver (joint solve, side dependencies): one word per node, kept only while
// the whole carve-up stays inside the LDS — beyond ~5,900 brokers the kernel runs without it
KAS_ABI_FN int32_t kas_order_wide_has_front(int32_t n_max) {
  int64_t n = n_max > 0 ? n_max : 1;
  return (int64_t)kas_order_wide_lds_core(n_max) + kas_align16(4 * (n + 1)) <= KAS_LDS_LIMIT ? 1 : 0;
}
// heat[] of the staging wave (class by node heat, KAS_WIDE_HEAT): uint16 per node, kept only while there is room
#ifndef KAS_WIDE_HEAT
#define KAS_WIDE_HEAT 4
#endif
KAS_ABI_FN int32_t kas_order_wide_has_heat(int32_t n_max) {
  int64_t n = n_max > 0 ? n_max : 1;
  if (KAS_WIDE_HEAT <= 0) return 0;
  return (int64_t)kas_order_wide_lds_core(n_max) + (kas_order_wide_has_front(n_max) ? kas_align16(4 * (n + 1)) : 0) +
         kas_align16(2 * (n + 1)) <= KAS_LDS_LIMIT ? 1 : 0;
}
KAS_ABI_FN int32_t kas_order_wide_lds(int32_t n_max) {
  int64_t n = n_max > 0 ? n_max : 1;
  return kas_order_wide_lds_core(n_max) + (kas_order_wide_has_front(n_max) ? kas_align16(4 * (n + 1)) : 0) +
         (kas_order_wide_has_heat(n_max) ? kas_align16(2 * (n + 1)) : 0);
}
#define KAS_ORDER_WIDE_BLOCK (64 * (3 + KAS_WIDE_BULK_SOLVERS))   // staging, retiring and the solver wavefronts
// round form of order: int32 count[n_max][CS] + uint64 lane masks [n_max]
// relaxation form of P5 (kas_order_relax.h), one wavefront per scenario: one uint32 counter word per node + the
// padding node's, 8 tag words, a row word per row and a staging word per (row, cell) pair of a tile (64 rows; the
// instance with double tiles: 128).  5,072 bytes at 1,000 brokers: four of these workgroups fit in the LDS that four
// workgroups of the fill kernel leave free on a CU.
// (with_ctx: the instance for batches with a Context keeps a second uint32 per node: what the rows add to count[n][2])
// (with_ids: the instances for int32 cells that keep the scenario's broker ids in the LDS too — kas_relax_lds_ids)
#ifndef KAS_RELAX_IDS_LDS_MAX
#define KAS_RELAX_IDS_LDS_MAX (64 * 1024)   // ... where the whole carve-up stays below this (about 7,900 brokers; 5,400 with a Context)
#endif
KAS_ABI_FN int32_t kas_order_relax_lds(int32_t n_max, int32_t double_tiles, int32_t with_ctx, int32_t with_ids = 0) {
  int64_t n = n_max > 0 ? n_max : 1;
  const int64_t rows = double_tiles >= 2 ? 256 : (double_tiles ? 128 : 64);   // (2: quad tiles, round 6)
  return kas_align16(kas_align16(4 * (n + 1)) + 8 * 4 + rows * 4 + 3 * rows * 4 + (with_ctx ? 4 * n : 0) + (with_ids ? 4 * n : 0));
}
// relaxation form for lists 4 and 5 wide (kas_order_relax_wide.h), one wavefront per scenario: a uint64 counter word per node + the
// padding node's, a row word per row and an 8-byte staging word per (row, cell) pair of a tile of 64 rows, the scenario's broker ids
KAS_ABI_FN int32_t kas_order_relaxw_lds(int32_t n_max, int32_t W) {
  int64_t n = n_max > 0 ? n_max : 1;
  return kas_align16(kas_align16(8 * (n + 1)) + 64 * 4 + 64 * (int64_t)W * 8 + 4 * n);
}
// ... launched where the plan says: KAS_PLAN_RELAX_TILES(1) at these widths, or — KAS_RELAXW_DEFAULT — whenever it applies
// (KAS_PLAN_TICKET_ORDER then names the wide ticket form)
#ifndef KAS_RELAXW_DEFAULT
#define KAS_RELAXW_DEFAULT 0
#endif
KAS_ABI_FN int32_t kas_relaxw_wanted(uint32_t flags) {
  if (flags & (KAS_FLAG_ROUND_ORDER | KAS_FLAG_TICKET_ORDER)) return 0;
  return (flags & KAS_FLAG_RELAX_TILES_64) != 0u || KAS_RELAXW_DEFAULT;
}
// Relaxation form on int32 cells: broker ids in the LDS (the IDL instances) for this broker count?  (KAS_TUNE_RELAX_GATHER_IDS:
// tuning builds that keep the gather from the L2-resident node table, for A/B)
KAS_ABI_FN int32_t kas_relax_lds_ids(int32_t n_max, int32_t with_ctx) {
#if defined(KAS_TUNE_RELAX_GATHER_IDS)
  (void)n_max; (void)with_ctx;
  return 0;
#else
  int64_t n = n_max > 0 ? n_max : 1;
  return kas_align16(kas_align16(4 * (n + 1)) + 8 * 4 + 128 * 4 + 3 * 128 * 4 + (with_ctx ? 4 * n : 0) + 4 * n) <= KAS_RELAX_IDS_LDS_MAX ? 1 : 0;
#endif
}
// Relaxation form: double tiles (128 rows, two rows per lane) in this launch?  A double tile halves the LDS round
// trips a scenario waits for (one batch of 1000 alone: order kernel 2.0 -> 1.7 ms) at ~1.2 x the LDS operations per row
// (3.97 evaluations per 128 rows against 3.26 per 64): right when the GPU is not full of wavefronts — few scenarios
// per launch — and wrong when it is (eight batches of 1000 in flight: 535k against 580k scenarios/s).
KAS_ABI_FN int32_t kas_relax_double_tiles(uint32_t flags, int32_t n_scenarios) {
  if (flags & KAS_FLAG_RELAX_TILES_128) return 1;
  if (flags & KAS_FLAG_RELAX_TILES_64) return 0;
  return n_scenarios < KAS_RELAX_DUAL_BELOW ? 1 : 0;
}
// Relaxation form: quad tiles (256 rows, four rows per lane; round 6) in a launch that takes double tiles on dword mid rows?
// Asked for by KAS_PLAN_RELAX_TILES(3) (both tile bits), or — KAS_RELAX_QUAD_BELOW — by batch size where neither bit is named.
// Where the instances do not apply (no dword mid rows, their LDS) such a launch keeps double tiles.
#ifndef KAS_RELAX_QUAD_BELOW
#define KAS_RELAX_QUAD_BELOW 0
#endif
KAS_ABI_FN int32_t kas_relax_quad_tiles(uint32_t flags, int32_t n_scenarios) {
  const uint32_t both = KAS_FLAG_RELAX_TILES_64 | KAS_FLAG_RELAX_TILES_128;
  if ((flags & both) == both) return 1;
  if (flags & both) return 0;
  return n_scenarios < KAS_RELAX_QUAD_BELOW ? 1 : 0;
}
// does a flag word (KAS_PLAN_* / KAS_FLAG_*) ask for the ticket form where the relaxation form is applicable?
KAS_ABI_FN int32_t kas_flags_want_tickets(uint32_t flags) {
  return (flags & (KAS_FLAG_TICKET_ORDER | KAS_FLAG_WIDE_COUNTERS)) != 0u || ((flags >> 12) & 0xfu) != 0u;
}
KAS_ABI_FN int32_t kas_order_round_lds(int32_t n_max, int32_t W) {
  int64_t n = n_max > 0 ? n_max : 1;
  return kas_align16(kas_align16(4 * n * kas_cnt_stride(W)) + 8 * n + 64);
}

// Hand-over words of a topic whose first fit is left to kas_p4_kernel (KasLaunch::p4s): [0] 1 = first fit pending (0: the
// fill kernel did it itself — the general fill — or the topic is not OK), [1] cap, [2 .. 2 + NW) orphans per chunk
#define KAS_P4S_HEAD 16
#define KAS_P4_WAVES 4
// LDS of kas_p4_kernel: load[n] int32, rack[n] int16, live[n] int16, control words
struct KasP4Lds { int32_t off_load, off_rack, off_live, off_ctl, total; };
KAS_ABI_FN KasP4Lds kas_p4_lds_layout(int32_t n_max) {
  KasP4Lds L;
  const int64_t n = n_max > 0 ? n_max : 1;
  int64_t o = 0;
  L.off_load = (int32_t)o; o = kas_align16(o + 4 * n);
  L.off_rack = (int32_t)o; o = kas_align16(o + 2 * n);
  L.off_live = (int32_t)o; o = kas_align16(o + 2 * n);
  L.off_ctl = (int32_t)o;  o = kas_align16(o + 4 * KAS_CTL_INTS);
  L.total = (int32_t)o;
  return L;
}

// LDS of kas_p4_order_kernel (first fit as a second wavefront of the relaxation form's workgroup, kas_order_relax.h): the order
// wavefront's carve-up, four 8-byte words between the two, first fit's carve-up
KAS_ABI_FN int32_t kas_p4_order_lds(int32_t n_max, int32_t double_tiles, int32_t with_ids) {
  return kas_align16(kas_order_relax_lds(n_max, double_tiles, 0, with_ids)) + 32 + kas_p4_lds_layout(n_max).total;
}
// First fit inside the order kernel's workgroup for this launch?  Lists up to 3 wide on the relaxation form without a Context and
// without the sampled verification, the fill kernel handing first fit over (what kas_split_p4 needs).  Asked for by naming BOTH
// first-fit switches (KAS_PLAN_SPLIT_P4 | KAS_PLAN_FILL_WITH_P4 = KAS_PLAN_P4_WITH_ORDER); by itself — KAS_P4_WITH_ORDER_BELOW —
// for batches too small to fill the GPU, where a scenario's latency is the batch's.
#ifndef KAS_P4_WITH_ORDER_BELOW
#define KAS_P4_WITH_ORDER_BELOW 512        // batches of fewer scenarios than this take it unless told otherwise (0: only on request) —
                                           // where kas_p4_kernel takes over (KAS_SPLIT_P4_FROM): 1000 scenarios alone 2.15 ms against 2.42 ms with
                                           // first fit inside the fill workgroup and 2.77 ms with kas_p4_kernel; twelve batches in flight 629k
                                           // scenarios/s against 753k with kas_p4_kernel (gpurun_out/r6h)
#endif
#define KAS_FLAG_P4_WITH_ORDER (KAS_FLAG_SPLIT_P4 | KAS_FLAG_FILL_WITH_P4)
// widths the kernels are instantiated for; a batch uses the smallest one >= its widest list
KAS_ABI_FN int32_t kas_width_class(int32_t W) { return W <= 2 ? 2 : W <= 5 ? W : 8; }

struct KasShape {
  int32_t W = 1;                      // widest list in the batch: max out_width
  int32_t Wc = 2;                     // instantiated kernel width class >= W
  int32_t n_max = 0;
  int32_t idmap_entries = 0;
  int32_t need_bsearch = 0;
  int32_t tickets_ok = 1;             // the ticket form of P5 is applicable to every scenario
  int32_t with_x = 1;                 // LDS has room for the histogram / quota table of the fast fill
  int32_t packed_ok = 1;              // every scenario's ticket bound fits 10-bit counter fields and no Context is handed in
  int32_t bound_small = 1;            // every scenario's ticket bound fits 10-bit counter fields
  int32_t any_ctx = 0;                // some scenario hands a Context in / wants it back
  int32_t wide_ok = 0;                // lists 4 or 5 wide and the wide ticket form is applicable
  int32_t relax_ok = 0;               // lists <= 3 wide and the relaxation form (kas_order_relax.h) is applicable to every scenario
  int32_t relaxw_ok = 0;              // lists 4 or 5 wide and the relaxation form for them (kas_order_relax_wide.h) is applicable
  int32_t bound_mid = 1;              // every scenario's ticket bound is below KAS_WIDE_COMMIT_LIMIT
  int32_t wide_checked = 0;           // wide_ok with a ticket bound of 1023 or more somewhere: the kernel checks its count
                                      // fields at the end and a scenario that outgrew them is solved again (fill + round form)
  int32_t round_fits = 1;             // the round form's LDS (int32 counters + 64-bit masks) fits 160 KiB
  int32_t fused_ok = 0;               // rack-diverse fill with per-chunk histograms (no chunk-count pass)
  KasLds lds_fused{};                 // its LDS carve-up (valid when fused_ok)
  int32_t max_partitions = 0;         // largest topic of the batch
  std::vector<int64_t> accmask_off;   // per scenario, in 64-bit words
  int64_t accmask_words = 0;
  std::vector<int64_t> orph_off;      // per scenario, in int32 elements
  int64_t orph_ints = 0;
  int32_t NW = 1;                     // wavefronts per scenario workgroup of the fill kernel
  int32_t G = 1;                      // lane groups (= scenarios) per wavefront, ticket form
  int64_t algorithmic_bytes = 0;     // int32 broker ids in and out, 4 P (cw + ow) per topic + 8 N per scenario (+ ctx)
  int64_t algorithmic_bytes16 = 0;   // the same for 16-bit cells (kas_plan_create16): 2 P (cw + ow) + 4 N (racks; no id table is read) (+ ctx)
  int64_t cur_need = 0, out_need = 0, aux_need = 0, ctx_need = 0;  // minimum pool lengths
  // first element of each pool a descriptor refers to (== *_need when none does): a batch that is a
  // slice of a larger one (kas_batch_slice) touches [lo, need) only, and the host path moves only that
  int64_t cur_lo = 0, out_lo = 0, aux_lo = 0, ctx_lo = 0;
  KasLds lds{};
};

// First fit in a kernel of its own for this launch?  The rack-diverse fill with four wavefronts per scenario, the
// one-workgroup fill kernel (the spread fill has its own P4 kernel).  `launch_flags`: KasLaunch::flags of the solve.
// By batch size unless the flags say (KAS_FLAG_SPLIT_P4 / KAS_FLAG_FILL_WITH_P4): one first-fit wavefront per scenario in a
// kernel of its own is what fills a GPU that has thousands of wavefronts to run (twelve batches of 1000 in flight: 672k
// against 617k scenarios/s); a batch that has the GPU to itself waits for that wavefront's windows one after the other
// (1000 scenarios alone: fill + first fit 1.39 ms against 1.07 ms on the fill's four wavefronts).
// (what either needs: the rack-diverse fill with four wavefronts per scenario, the one-workgroup fill kernel, first fit's LDS)
static inline bool kas_p4_handover_ok(const KasShape& s, int32_t nw, uint32_t launch_flags, int32_t spread_chunks) {
  return s.with_x && nw == KAS_P4_WAVES && !(launch_flags & KAS_FLAG_GENERIC_FILL) && spread_chunks == 0 &&
         kas_p4_lds_layout(s.n_max).total <= KAS_LDS_LIMIT;
}
// first fit in kas_p4_kernel: KAS_PLAN_SPLIT_P4 by itself, or — neither switch named — batches of >= KAS_SPLIT_P4_FROM scenarios
static inline bool kas_split_p4(const KasShape& s, int32_t nw, uint32_t launch_flags, int32_t spread_chunks, int32_t n_scenarios) {
  if (!kas_p4_handover_ok(s, nw, launch_flags, spread_chunks)) return false;
  const uint32_t sw = launch_flags & KAS_FLAG_P4_WITH_ORDER;
  return sw == KAS_FLAG_SPLIT_P4 || (sw == 0u && n_scenarios >= KAS_SPLIT_P4_FROM);
}
// first fit inside the order kernel's workgroup (kas_p4_order_kernel): both switches named, or — neither — batches of fewer than
// KAS_P4_WITH_ORDER_BELOW scenarios; where the kernel does not apply such a launch keeps first fit inside the FILL workgroup.
// `relax_plain`: the launch takes the relaxation form without a Context and without the sampled verification, in an instance that
// waits for no gather; `with_ids`: that instance keeps the broker ids in the LDS
static inline bool kas_p4_with_order(const KasShape& s, int32_t nw, uint32_t launch_flags, int32_t spread_chunks, int32_t n_scenarios,
                                     bool relax_plain, int32_t double_tiles, int32_t with_ids) {
  if (!relax_plain || !kas_p4_handover_ok(s, nw, launch_flags, spread_chunks)) return false;
  if (kas_p4_order_lds(s.n_max, double_tiles, with_ids) > KAS_LDS_LIMIT) return false;
  const uint32_t sw = launch_flags & KAS_FLAG_P4_WITH_ORDER;
  return sw == KAS_FLAG_P4_WITH_ORDER || (sw == 0u && n_scenarios < KAS_P4_WITH_ORDER_BELOW);
}

// Dword mid rows (KAS_FLAG_MID32) in this launch?  int32 cells, the batch's lists 3 wide at most and that width class, at most
// KAS_MID32_N_MAX brokers, the relaxation form in the instances with the broker ids in the LDS, no Context, no sampled verification
// (`launch_flags` >> 24), no index rows (their layout is the 16-bit row's), no spread fill (its kernels write 16-bit rows).
// `user_flags`: the word of kas_plan_set_flags / the emulator (KAS_PLAN_MID32 / KAS_PLAN_NO_MID32).
static inline bool kas_mid32_launch(const KasShape& s, bool c16, uint32_t user_flags, bool relax, uint32_t launch_flags, int32_t with_ids,
                                    bool index_rows, int32_t spread_chunks) {
  return kas_mid32_wanted(user_flags) && !c16 && s.Wc == 3 && s.n_max <= KAS_MID32_N_MAX && relax && !s.any_ctx &&
         (launch_flags >> 24) == 0u && with_ids && !index_rows && spread_chunks == 0;
}

// Fused histogram layout of the rack-diverse fill (kas_fill_lds_layout with_x = 2): lists up to 3 wide
// (the instantiated variants), more than one chunk, a chunk's rows countable in uint16, and the larger
// table must not cost a resident workgroup (4 per CU is what the kernel's 128 VGPRs allow anyway).
static inline void kas_choose_fused(KasShape* s) {
  s->fused_ok = 0;
  if (!s->with_x || s->Wc > 3 || s->NW < 2) return;
  const int64_t tiles = ((int64_t)s->max_partitions + 63) / 64;
  if ((tiles / s->NW + 2) * 64 > 65535) return;
  const KasLds f = kas_fill_lds_layout(s->n_max, s->Wc, s->NW, s->idmap_entries, s->need_bsearch, 2);
  if (f.total > KAS_LDS_LIMIT) return;
  const int per_cu_fused = KAS_LDS_LIMIT / f.total, per_cu_now = KAS_LDS_LIMIT / s->lds.total;
  if (per_cu_fused < (per_cu_now < 4 ? per_cu_now : 4)) return;
  s->fused_ok = 1;
  s->lds_fused = f;
}

// Spread fill: chunks per scenario, or 0 when the batch takes the one-workgroup fill kernel.  For
// batches of few, large, single-topic scenarios (BASELINE.json configs[4]: 1M rows — one workgroup
// streams them in 10 ms, 256 one-wavefront workgroups in a fraction of that); `force` (KAS_FLAG_SPREAD_FILL)
// takes any single-topic batch, with few chunks, so that small cases exercise the same kernels.
static inline bool kas_batch_single_topic(const kas_batch_desc* b) {
  if (b->n_scenarios <= 0 || b->n_topics != b->n_scenarios) return false;
  for (int32_t i = 0; i < b->n_scenarios; ++i) if (b->scenarios[i].topic_count != 1) return false;
  return true;
}
static inline int32_t kas_spread_chunks(const KasShape& s, int32_t n_scenarios, bool single_topic, bool force) {
  if (!s.with_x || n_scenarios <= 0 || !single_topic) return 0;
  // the layout of a one-wavefront workgroup (histogram rows, node tables) must fit as well
  if (kas_fill_lds_layout(s.n_max, s.Wc, 1, s.idmap_entries, s.need_bsearch, 1).total > KAS_LDS_LIMIT) return 0;
  const int64_t tiles = ((int64_t)s.max_partitions + 63) / 64;
  // (pass A counts a chunk's rows per node and sweep in uint16 cells: a chunk stays below 1,023 tiles)
  if (force) return (int32_t)(tiles >= 6 * 1023 ? 0 : (tiles >= 6 ? 6 : (tiles > 0 ? tiles : 1)));
  if (n_scenarios > 64 || tiles < 2048) return 0;
  int64_t ch = 512 / n_scenarios;
  if (ch > 256) ch = 256;
  while (ch > 1 && tiles / ch < 16) ch >>= 1;
  while ((tiles + ch - 1) / ch >= 1023) ch <<= 1;          // (uint16 cells: more, shorter chunks)
  return (int32_t)(ch >= 4 ? ch : 0);
}

// Validate descriptors and derive everything a launch needs.  Returns KAS_E_* and fills err.
static inline int kas_shape_batch(const kas_batch_desc* b, KasShape* sh, std::string* err,
                                  int want_waves = 0, int want_groups = 0) {
  auto fail = [&](int code, const std::string& m) { if (err) *err = m; return code; };
  if (want_waves != 0 && want_waves != 1 && want_waves != 2 && want_waves != 4 && want_waves != 8)
    return fail(KAS_E_INVALID_ARG, "waves per scenario must be 1, 2, 4 or 8");
  if (want_groups != 0 && want_groups != 1 && want_groups != 2 && want_groups != 4)
    return fail(KAS_E_INVALID_ARG, "scenarios per wavefront must be 1, 2 or 4");
  if (!b || b->n_scenarios < 0 || b->n_topics < 0) return fail(KAS_E_INVALID_ARG, "null/negative batch");
  if (b->n_scenarios > 0 && (!b->scenarios)) return fail(KAS_E_INVALID_ARG, "scenarios == NULL");
  if (b->n_topics > 0 && !b->topics) return fail(KAS_E_INVALID_ARG, "topics == NULL");
  KasShape s;
  int32_t relax_inputs_ok = 1;         // no KAS:190 index error, no int overflow, rows per node inside the relaxation form's fields
  const int64_t kNone = INT64_MAX;
  s.cur_lo = s.out_lo = s.aux_lo = s.ctx_lo = kNone;
  s.accmask_off.assign((size_t)b->n_scenarios, 0);
  s.orph_off.assign((size_t)b->n_scenarios, 0);
  int64_t max_range_fit = 0;
  for (int32_t i = 0; i < b->n_scenarios; ++i) {
    const kas_scenario_desc& sd = b->scenarios[i];
    if (sd.n_nodes < 0 || sd.topic_count < 0 || sd.topic_begin < 0 ||
        (int64_t)sd.topic_begin + sd.topic_count > b->n_topics)
      return fail(KAS_E_INVALID_ARG, "scenario " + std::to_string(i) + ": bad topic range / n_nodes");
    if (sd.n_nodes > KAS_N_LIMIT)
      return fail(KAS_E_UNSUPPORTED, "scenario " + std::to_string(i) + ": more than 32767 brokers");
    if (sd.n_nodes > 0) {
      if (sd.node_off < 0 || sd.node_off + sd.n_nodes > b->node_pool_len || !b->node_id || !b->node_rack)
        return fail(KAS_E_INVALID_ARG, "scenario " + std::to_string(i) + ": node_off outside the node pool");
      int64_t lo = b->node_id[sd.node_off], hi = b->node_id[sd.node_off + sd.n_nodes - 1];
      int64_t range = hi - lo + 1;
      if (range >= 1 && range <= KAS_IDMAP_CAP) { if (range > max_range_fit) max_range_fit = range; }
      else s.need_bsearch = 1;      // sparse ids (or unsorted: the kernel reports BAD_NODES)
    }
    // Context handed in (KAS:360-369).  The ticket forms seed their count fields from it and write
    // them back (tickets and commit counts start at zero in every solve, so only the FIELD width
    // matters): the kernel checks "largest counter + rows a node can gain < field limit" per scenario
    // and leaves a scenario that fails it to the round form (ord_flag), which therefore has to fit.
    // 4 x uint16 counter rows: the packed 3 x 10-bit row keeps the commits as the sum of its counts.
    if (sd.ctx_off >= 0) {
      s.any_ctx = 1;
      if (sd.ctx_width < 1 || sd.ctx_width > KAS_MAX_WIDTH)
        return fail(KAS_E_INVALID_ARG, "scenario " + std::to_string(i) + ": ctx_width outside [1,8]");
      int64_t e = sd.ctx_off + (int64_t)sd.n_nodes * sd.ctx_width;
      if (e > s.ctx_need) s.ctx_need = e;
      if (sd.n_nodes > 0 && sd.ctx_off < s.ctx_lo) s.ctx_lo = sd.ctx_off;
      s.algorithmic_bytes += 8ll * sd.n_nodes * sd.ctx_width;
      s.algorithmic_bytes16 += 8ll * sd.n_nodes * sd.ctx_width;
    }
    if (sd.n_nodes > s.n_max) s.n_max = sd.n_nodes;
    s.algorithmic_bytes += 8ll * sd.n_nodes;
    s.algorithmic_bytes16 += 4ll * sd.n_nodes;
    int64_t words = 0, rows = 0, rows_sum = 0, ticket_bound = 0;
    for (int32_t k = 0; k < sd.topic_count; ++k) {
      const kas_topic_desc& td = b->topics[sd.topic_begin + k];
      std::string where = "scenario " + std::to_string(i) + " topic " + std::to_string(k) + ": ";
      if (td.n_partitions < 0) return fail(KAS_E_INVALID_ARG, where + "negative n_partitions");
      if (td.cur_width < 0 || td.cur_width > KAS_MAX_WIDTH)
        return fail(KAS_E_UNSUPPORTED, where + "cur_width outside [0,8]");
      if (td.out_width < 1 || td.out_width > KAS_MAX_WIDTH || td.out_width < td.cur_width)
        return fail(KAS_E_UNSUPPORTED, where + "out_width outside [max(1,cur_width),8]");
      if (td.rf >= 1 && td.rf <= sd.n_nodes && td.rf > td.out_width)
        return fail(KAS_E_UNSUPPORTED, where + "rf exceeds out_width");
      if (td.cur_off < 0 || td.out_off < 0) return fail(KAS_E_INVALID_ARG, where + "negative pool offset");
      if (td.out_width > s.W) s.W = td.out_width;
      if (td.n_partitions > s.max_partitions) s.max_partitions = td.n_partitions;
      int64_t P = td.n_partitions;
      int64_t ce = td.cur_off + P * td.cur_width, oe = td.out_off + P * td.out_width;
      if (ce > s.cur_need) s.cur_need = ce;
      if (oe > s.out_need) s.out_need = oe;
      if (ce > td.cur_off && td.cur_off < s.cur_lo) s.cur_lo = td.cur_off;
      if (oe > td.out_off && td.out_off < s.out_lo) s.out_lo = td.out_off;
      const int64_t aux_offs[3] = {td.cur_len_off, td.in_partitions_off, td.part_id_off};
      for (int64_t ao : aux_offs) {
        if (ao < -1) return fail(KAS_E_INVALID_ARG, where + "aux offset < -1");
        if (ao >= 0 && ao + P > s.aux_need) s.aux_need = ao + P;
        if (ao >= 0 && P > 0 && ao < s.aux_lo) s.aux_lo = ao;
      }
      int64_t w = (int64_t)td.cur_width * ((P + 63) / 64);
      if (w > words) words = w;
      if (((P + 63) / 64) * 64 > rows) rows = ((P + 63) / 64) * 64;
      rows_sum += ((P + 63) / 64) * 64;
      s.algorithmic_bytes += 4ll * P * (td.cur_width + td.out_width);
      s.algorithmic_bytes16 += 2ll * P * (td.cur_width + td.out_width);
      // a node never holds more than cap rows of a topic (KAS:65-71, cap over <= P partitions), and
      // a ticket on node n counts the rows that hold n so far in the scenario
      if (td.rf >= 1 && td.rf <= sd.n_nodes && sd.n_nodes > 0) {
        const int64_t prod = P * (int64_t)td.rf;
        if (prod >= (1ll << 31)) { s.tickets_ok = 0; relax_inputs_ok = 0; }
        ticket_bound += (prod + sd.n_nodes - 1) / sd.n_nodes;
      }
      if (td.name_hash == (int32_t)0x80000000) { s.tickets_ok = 0; relax_inputs_ok = 0; }   // KAS:190 index error: round form
    }
    if (ticket_bound >= KAS_TICKET_LIMIT) s.tickets_ok = 0;
    if (ticket_bound >= KAS_RELAX_ROW_LIMIT) relax_inputs_ok = 0;
    if (ticket_bound >= KAS_PACKED_TICKET_LIMIT) s.bound_small = 0;
    if (ticket_bound >= KAS_WIDE_COMMIT_LIMIT) s.bound_mid = 0;
    s.accmask_off[(size_t)i] = s.accmask_words;
    s.accmask_words += words > 0 ? words : 1;
    // (every topic its own stretch of the scenario's orphan scratch: with the first fit in a kernel of its own a topic's list
    // is read after the fill kernel has gone on to the next topic)
    s.orph_off[(size_t)i] = s.orph_ints;
    s.orph_ints += rows_sum > 0 ? rows_sum : 64;
  }
  if (s.cur_lo == kNone || s.cur_lo > s.cur_need) s.cur_lo = s.cur_need;
  if (s.out_lo == kNone || s.out_lo > s.out_need) s.out_lo = s.out_need;
  if (s.aux_lo == kNone || s.aux_lo > s.aux_need) s.aux_lo = s.aux_need;
  if (s.ctx_lo == kNone || s.ctx_lo > s.ctx_need) s.ctx_lo = s.ctx_need;
  s.idmap_entries = (int32_t)max_range_fit;
  s.Wc = kas_width_class(s.W);
  // lists 4 and 5 wide: the wide ticket form (kas_order_wide.h) under the same conditions plus 10-bit
  // count fields and 16-bit LDS offsets of its 8-byte counter rows; beyond 5: round form
  s.packed_ok = s.bound_small && !s.any_ctx;
  // relaxation form: no KAS:190 index error, rows per node inside its 16-bit count fields (65,535, where the ticket
  // forms' 16-bit tickets end too; the packed ticket form stops at 1,023) — and none of the ticket form's 16-bit LDS offsets, so
  // the broker count is limited by the fill kernel's LDS only.  A Context handed in is checked per scenario by the
  // kernel (its counters + the rows to come must fit the fields; else the round form, which fits whenever a batch
  // with a Context is accepted at all)
  s.relax_ok = s.Wc <= 3 && relax_inputs_ok && kas_order_relax_lds(s.n_max, 1, s.any_ctx, kas_relax_lds_ids(s.n_max, s.any_ctx)) <= KAS_LDS_LIMIT;
  // ... and at lists 4 and 5 wide: no Context handed in, the broker ids in the LDS beside the 8-byte counter words
  s.relaxw_ok = (s.Wc == 4 || s.Wc == 5) && relax_inputs_ok && !s.any_ctx && kas_order_relaxw_lds(s.n_max, s.Wc) <= 96 * 1024;
  // (a node that may hold 1023 .. 2039 rows: the count fields are checked after the fact, and what outgrew them goes
  // to the round form — which must then fit)
  s.wide_ok = (s.Wc == 4 || s.Wc == 5) && s.tickets_ok && s.bound_mid && 8 * ((int64_t)s.n_max + 1) <= 65536 &&
              kas_order_wide_lds(s.n_max) <= KAS_LDS_LIMIT &&
              (s.bound_small || kas_order_round_lds(s.n_max, s.Wc) <= KAS_LDS_LIMIT);
  s.wide_checked = s.wide_ok && !s.bound_small;
  if (s.Wc > 3) s.tickets_ok = 0;              // ring slots / packed counter rows hold lists up to 3
  // widest fill workgroup whose LDS carve-up fits: 4 wavefronts per scenario by default
  int err_total = 0;
  for (int nw = want_waves > 0 ? want_waves : 4; nw >= 1; nw >>= 1) {
    KasLds l = kas_fill_lds_layout(s.n_max, s.Wc, nw, s.idmap_entries, s.need_bsearch, 1);
    err_total = l.total;
    if (l.total <= KAS_LDS_LIMIT) { s.lds = l; s.NW = nw; err_total = 0; break; }
  }
  if (err_total) {
    // many brokers x wide lists (BASELINE configs[4]: 5k brokers, RF 5): no room for the
    // histogram, the general sticky fill needs none
    KasLds l = kas_fill_lds_layout(s.n_max, s.Wc, 1, s.idmap_entries, s.need_bsearch, 0);
    err_total = l.total;
    if (l.total <= KAS_LDS_LIMIT) { s.lds = l; s.NW = 1; s.with_x = 0; err_total = 0; }
  }
  kas_choose_fused(&s);
  // two scenarios per solver wavefront by default: a scenario rarely has more than ~20 rows ready
  // at once, so 32 lanes serve it as well as 64 and the wave's instructions are shared (even a
  // lone scenario is no faster on 64 lanes: its stager then tickets each tile in two halves).
  // Counter rows are addressed with 16-bit LDS byte offsets: fewer scenarios per wavefront when the
  // groups' regions do not fit 64 KiB, no ticket form at all when one region does not — decided
  // BEFORE the fallback check below, so that a plan no order kernel can serve is refused here and
  // not at its first launch.
  // (what is addressed that way are the counter rows, which come first in a group's region: with ONE group they
  // have to end below 64 KiB, not the lane masks and ticket counts behind them — round 3: one group serves up to
  // 8,191 brokers instead of 4,680.  Several groups keep the old rule, every region below 64 KiB: where two
  // regions no longer fit that, one scenario per wavefront at three workgroups per CU is the better plan anyway.)
  s.G = want_groups > 0 ? want_groups : 2;
  while (s.G > 1 && (int64_t)s.G * kas_order_ticket_group_bytes(s.n_max, s.G, 0) > 65536) s.G >>= 1;
  if (s.G == 1 && (8 * ((int64_t)s.n_max + 1) > 65536 || kas_order_ticket_lds(s.n_max, 1, 0) > KAS_LDS_LIMIT)) s.tickets_ok = 0;
  // the round form of P5 is the universal fallback; a batch that one of the ticket forms serves does
  // not need it to fit (KAS_PLAN_ROUND_ORDER is refused for such a plan, see round_fits)
  s.round_fits = kas_order_round_lds(s.n_max, s.Wc) <= KAS_LDS_LIMIT;
  if (err_total == 0 && !s.round_fits && (!(s.tickets_ok || s.wide_ok || s.relax_ok) || s.any_ctx))
    err_total = kas_order_round_lds(s.n_max, s.Wc);
  if (err_total)
    return fail(KAS_E_UNSUPPORTED, "broker count " + std::to_string(s.n_max) + " x width " +
                std::to_string(s.Wc) + " needs " + std::to_string(err_total) +
                " B of LDS (limit 163840)");
  *sh = s;
  return KAS_E_OK;
}
