// kas_order_wide.h — P5 (computePreferenceLists, KAS:202-239), ticket form for replica lists 4 and 5
// wide (BASELINE.json configs[4]: 1M partitions x 5k brokers, RF 5).  Included by kas_solver_body.h.
//
// Same plan as order_tickets<W <= 3> — a STAGING wavefront (mid rows from HBM, tickets, ring slots), a
// RETIRING wavefront (node index -> broker id, digest, final out row), talking only through the tags of
// a ring of K tiles of 64 slots — but with THREE solver wavefronts, one per SIMD the other two leave
// free: a single wavefront issues one instruction of its dependent chain every ~5 cycles, so the
// solver, not the dependency chain, was what a 1M-row scenario waited for.  The staging wave sorts
// every row into one of two classes and appends its slot to that class's claim list:
//   class 1 (wave 1): rows holding a node that >= KAS_WIDE_CHAIN_DENSITY rows of the same 64-row
//     tile hold — the brokers that first fit is filling with consecutive orphans.  All of this
//     wave's 64 rows in hand sit on those chains, so one joint step decides many of them at once.
//   class 0 (waves 3, 4): everything else — rows whose nodes are ~N/W rows apart, almost always ready.
// Tickets make any split exact: a row commits when every earlier row on each of its nodes has
// (ticket == commits of the node), whichever wave holds them.  No deadlock: each solver claims its
// class in row order, so the oldest undecided row of the scenario is always in some lane's hand,
// and its tickets are all due.  One scenario per workgroup.
//
// What differs from the 3-wide kernel:
//   * count[n][0..5) live in one uint64 per node as five count fields (bits 0, 10, 20, 32, 42) read as 10-bit
//     values, plus the commits on n (their sum) in the 11 bits from bit 53.  Safe a priori while no node can
//     hold 1023 rows of the scenario (the plan checks the bound; configs[4] has cap 981).  Up to 2039 rows per
//     node (round 3) the kernel runs all the same and CHECKS: a count that outgrows 10 bits carries into the
//     next field — or, for fields 2 and 4, which have 12 and 11 bits of room, grows past 1023 in place — so
//     when the last row has retired "the five fields add up to the commits, and field 2 is below 1024" holds
//     for every node exactly if no count was ever misread (counts only grow, and the counts a row of a joint
//     step sees are counts the node really reaches).  The commits themselves cannot be touched: field 4 would
//     have to pass 2047.  So a scenario that fails the check was solved on wrong counts but to the end (every
//     pick is one of the row's holders whatever the counts, the relaxation rounds end with the dependency
//     depth), it is flagged (KasLaunch::ord_flag) and solved again — fill and round form — behind this kernel.
//   * holders are stored ascending (Sets.newTreeSet, KAS:228) and the picks are the general
//     "minimum of (count, visit position) over the nodes still in the set" of pick_row<W>
//     (KAS:263-278), five times; the rotation offsets idx_m = abs(hash) % m travel in the slot.
//   * the joint solve (KAS_WIDE_JOINT).  First fit hands consecutive orphans to one node
//     (at configs[4] each added broker takes ~981 of them: the P5 dependency chain is ~180k rows long
//     against 15.6k tiles), rack constraints keep two or three such nodes open at once, and two rows
//     of a hand often share an old broker.  A solver step therefore decides every row in hand whose
//     dependencies are in the hand too: rows deep in a queue vote for the KAS_WIDE_HOT nodes that get a
//     rank layout (rank = ticket - commits), a row may also wait on one more node with a single row
//     ahead if that row is in the hand (front[]), the set is closed under "every row ahead of me on my
//     named nodes, and my side row, is in the set", and the picks are relaxed against per-node prefix
//     sums of the wins (ballot + v_mbcnt over the rank layouts, added to the packed counter words)
//     until nothing changes: the sequential answer, because the set is closed and the dependencies
//     acyclic.  (Round 2 also carried the single-node threshold queues of the 3-wide kernel, generalised to
//     five picks, as a build switch: 57.0 ms against 39.6 ms at configs[4]; removed in round 3, git has it.)
#pragma once

namespace kas {

// rows a joint step must decide beyond the ones that were ready anyway for it to count as paying;
// below that the next attempts are skipped (1, 2, 4 ... up to KAS_WIDE_BACKOFF_MAX steps)
#ifndef KAS_WIDE_MIN_GAIN
#define KAS_WIDE_MIN_GAIN 1
#endif
#ifndef KAS_WIDE_BACKOFF_MAX
#define KAS_WIDE_BACKOFF_MAX 0
#endif
#ifndef KAS_WIDE_RING_SLOTS
#define KAS_WIDE_RING_SLOTS 8
#endif
// rows of one staged tile on the same node from which the node counts as "being filled" (class 1)
#ifndef KAS_WIDE_CHAIN_DENSITY
#define KAS_WIDE_CHAIN_DENSITY 2
#endif
// wavefronts of the workgroup: 0 stages, 1 solves class 1, 2 retires, 3 .. 3 + KAS_WIDE_BULK_SOLVERS - 1
// solve class 0 (solver b takes the list entries b, b + NB, ...: in row order each, no claim races).
// A workgroup's waves go round the four SIMDs, so wave 4 shares its SIMD with the staging wave, not
// with the class-1 solver.
#define KAS_WIDE_WAVES (3 + KAS_WIDE_BULK_SOLVERS)
#define KAS_WIDE_STAGER 0
#define KAS_WIDE_CHAIN_SOLVER 1
#define KAS_WIDE_RETIRER 2
// steps without a joint attempt after one that paid (0: none)
#ifndef KAS_WIDE_SKIP_AFTER_PASS
#define KAS_WIDE_SKIP_AFTER_PASS 0
#endif
// Joint solve: every row in hand whose pending holders are all among KAS_WIDE_HOT named nodes — and whose
// predecessors on those nodes are in hand and decidable too — is decided in one step, rows that sit in TWO
// queues included.  Naming: every row deep in a queue names its deepest one, and the most named nodes among
// the first KAS_WIDE_HOT + KAS_WIDE_VOTE names are taken.
#ifndef KAS_WIDE_VOTE
#define KAS_WIDE_VOTE 1
#endif
// a row votes (and a joint step is tried at all) when it has more than this many rows ahead of it on some
// node (configs[4], order kernel: 0 -> 46.2 ms, 1 -> 41.2, 2 -> 40.3, 3 -> 39.6, 4 -> 40.2, 6 -> 41.8)
#ifndef KAS_WIDE_VOTE_DEPTH
#define KAS_WIDE_VOTE_DEPTH 3
#endif
// side dependencies of the joint solve (0: rows that wait on a node that is not named are left out)
#ifndef KAS_WIDE_SIDE
#define KAS_WIDE_SIDE 1
#endif
#define KAS_WIDE_FIELD_MASK 0x3ffu
#define KAS_WIDE_COMMIT_SHIFT 53                          // commits of a node: 11 bits from here (bit 21 of the high word)
#define KAS_WIDE_DUMMY_TICKET 0x7ff                      // the padding holder's ticket == its row's commits field

struct alignas(16) WideSlot { int32_t tag; int32_t e[5]; int32_t rot; int32_t spare; };
// tag: KAS_TAG_FREE / KAS_TAG_END as in the 3-wide kernel
//      staged:  row counter j of the lane (bits 0..25) | list length Lp << 26 (0..5)
//      done:    0x80000000 | pos[0] | pos[1] << 3 | ... | pos[4] << 12 | Lp << 15
#define KAS_WTAG_DONE ((int32_t)0x80000000)
#define KAS_WTAG_IS_DONE(t) (((uint32_t)(t) & 0xfffc0000u) == 0x80000000u)

// slot.rot = TileIter::rot of a wide iterator: idx_m = Math.abs(hash) % m (KAS:190) for set sizes m = 1..5, 3 bits each
// at bit 3 m (computed per topic by tile_next_topic)

// counter row of a node: five counts at bits 0, 10, 20, 32, 42 (read as 10-bit fields; field 2 has room up to
// bit 31, field 4 up to bit 52) and the number of rows that committed on the node (= their sum, kept
// separately so that readiness is one shift) in the 11 bits from bit 53 (a node holds fewer than
// KAS_WIDE_COMMIT_LIMIT rows of the scenario: the plan checks)
KAS_DEV uint64_t wide_field_unit(int r) {                   // + 1 on count field r and on the commits
  return (1ull << KAS_WIDE_COMMIT_SHIFT) +
         (r == 0 ? 1ull : r == 1 ? (1ull << 10) : r == 2 ? (1ull << 20) : r == 3 ? (1ull << 32) : (1ull << 42));
}

// The picks of one row (KAS:225-236) as pick_row<W>, with the rotation offsets taken straight from
// the packed word (idx_m at bit 3 m) and the ranks of the set members carried along instead of
// recounted.  c[k][r] = count[holder k][r], holders ascending; pos[r] = list position picked for r.
template <int W>
KAS_DEV void pick_row_packed(const int32_t (&c)[W][W], int32_t Lp, bool valid, int32_t rot, int32_t (&pos)[W]) {
  // Straight-line arithmetic, no compare / select pairs (on gfx950 each costs a wait state besides its two issue
  // slots, and this function is most of a relaxation round of the chain solver's joint step):
  //   alive4   nibble k = 8 while holder k is still to be picked
  //   t        8 * (rank of the next set member + idx) + k, so that (t mod 8 m) is visit rank << 3 | list position
  //   key      count << 6 | visit rank << 3 | list position, bit 30 set for a holder that is not alive: the minimum
  //            names its own position (a dead row picks some position < W; nothing reads it)
  uint32_t alive4 = valid ? (0x88888888u & ((1u << (4 * Lp)) - 1u)) : 0u;
  int32_t m = valid ? Lp : 0;
#pragma unroll
  for (int r = 0; r < W; ++r) {
    if (r == W - 1) {                                       // at most one node is left: nothing to compare
      int32_t ps = W - 1;
#pragma unroll
      for (int k = W - 2; k >= 0; --k) ps = ((alive4 >> (4 * k + 3)) & 1u) ? k : ps;
      pos[r] = ps;
      break;
    }
    const uint32_t m8 = (uint32_t)m << 3;
    uint32_t t = (uint32_t)((rot >> (3 * m)) & 7) << 3;
    uint32_t best = 0x7fffffffu;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const uint32_t in8 = (alive4 >> (4 * k)) & 0xfu;      // 8 or 0
      const uint32_t wrapped = t - m8;                      // (huge when t < 8 m)
      const uint32_t rr8k = wrapped < t ? wrapped : t;
      const uint32_t key = ((uint32_t)c[k][r] << 6) | rr8k | ((in8 ^ 8u) << 27);
      best = key < best ? key : best;
      t += in8 + 1u;
    }
    const int32_t ps = (int32_t)(best & 7u);
    pos[r] = ps;
    alive4 &= ~(0xfu << (4 * ps));                          // nodeSet.remove (KAS:232)
    m -= m > 0 ? 1 : 0;
  }
}

template <int W>
KAS_DEV void order_tickets_wide(const KasLaunch& a, int32_t s_index, unsigned char* lds_raw) {
  static_assert(W == 4 || W == 5, "the narrow kernel serves lists up to 3 wide, the round form beyond 5");
  constexpr int K = KAS_WIDE_RING_SLOTS;
  constexpr int T = W - 1;                                  // replica indices whose counts decide a pick
  const int lane = kasw::lane();
  const int32_t wave = kasw::wave_id();
  const bool have_s0 = s_index < a.n_scenarios;
  const int32_t s = have_s0 ? s_index : a.n_scenarios;
  const int32_t nmax = a.n_max > 0 ? a.n_max : 1;
  uint64_t* cnt = (uint64_t*)lds_raw;                       // [nmax + 1]: + the padding holder's row
  uint64_t* dep = (uint64_t*)(lds_raw + kas_align16(8 * (int64_t)(nmax + 1)));                  // lane mask per node
  uint16_t* run = (uint16_t*)((unsigned char*)dep + kas_align16(8 * (int64_t)(nmax + 1)));       // tickets handed out
  WideSlot* ring = (WideSlot*)((unsigned char*)run + kas_align16(2 * (int64_t)(nmax + 1)));
  uint16_t* clist = (uint16_t*)(ring + K * 64);             // [2][K * 64] claim lists: ring slot of the class's next rows
  uint64_t* gdig = (uint64_t*)(clist + 2 * K * 64);
  uint32_t* lstate = (uint32_t*)(gdig + 1);                 // [2] rows appended to each list | 1 << 31 once staging has ended
  uint32_t* rank_owner_all = (uint32_t*)(gdig + 2);         // [1 + NB][KAS_WIDE_HOT][64] queue scratch of the solvers: rank -> lane
  uint32_t* wd = rank_owner_all + 64 * KAS_WIDE_HOT * (1 + KAS_WIDE_BULK_SOLVERS);       // watchdog word (see watchdog_poll)
  // front[n] = the class-1 solver's row in hand that is next to commit on node n: step stamp << 11 | the
  // node's position in that row's list << 8 | lane (joint solve, side dependencies)
  uint32_t* front = wd + 4;                                 // [nmax + 1], if the LDS has room for it
  const bool has_front = kas_order_wide_has_front(a.n_max) != 0;
  // heat[n] = the last tile in which >= KAS_WIDE_CHAIN_DENSITY rows held node n (staging wave only): a node
  // first fit is filling stays "hot" for KAS_WIDE_HEAT tiles, and rows holding a hot node go to the chain
  // solver even when they are the only such row of their tile (rack conflicts send part of a run of
  // orphans to the next brokers, which then see one row per tile or fewer: sorted by tile density alone
  // those rows land in a bulk solver's hand and every one of them cuts the chain solver's queue in two)
  uint16_t* heat = (uint16_t*)((unsigned char*)front + (has_front ? kas_align16(4 * (int64_t)(nmax + 1)) : 0));
  const bool has_heat = kas_order_wide_has_heat(a.n_max) != 0;
  // padding holder: a ticket that always matches its commits; never picked (pick_row looks at Lp cells)
  const int32_t dummy_e = (KAS_WIDE_DUMMY_TICKET << 16) | (nmax * 8);

  kas_scenario_desc sd;
  sd.n_nodes = 0; sd.topic_begin = 0; sd.topic_count = 0; sd.ctx_width = 0; sd.node_off = 0; sd.ctx_off = -1;
  if (have_s0) sd = a.scen[s];
  const int32_t* g_node_id = a.node_id + sd.node_off;
  constexpr int NB = KAS_WIDE_BULK_SOLVERS;
  // Context handed in (kas_solver_body.h, ctx_gain_bound): it seeds the 10-bit count fields; a scenario
  // whose counters would not stay below 1023 is left to the round form.  (workgroup-uniform)
  bool have_s = have_s0;
  int32_t* g_ctx = nullptr;
  int32_t ccols = 0;
  if (have_s0 && sd.ctx_off >= 0 && sd.ctx_width > 0 && a.ctx != nullptr) {
    g_ctx = a.ctx + sd.ctx_off;
    ccols = sd.ctx_width < W ? sd.ctx_width : W;
    const bool over = ctx_over_limit(g_ctx, sd.n_nodes, sd.ctx_width, ccols, (uint32_t)KAS_PACKED_TICKET_LIMIT,
                                     ctx_gain_bound(a, sd), lane, 64);
    if (kasw::ballot(over) != 0ull) {
      have_s = false; g_ctx = nullptr; ccols = 0;
      if (wave == 0 && lane == 0 && a.ord_flag) a.ord_flag[s] = 1;
    }
  }
  for (int32_t n = lane + 64 * wave; n <= nmax; n += 64 * KAS_WIDE_WAVES) {
    uint64_t x = 0ull;
    if (n < sd.n_nodes)
      for (int32_t r = 0; r < ccols; ++r)
        x |= (uint64_t)(uint32_t)g_ctx[(int64_t)n * sd.ctx_width + r] << (r < 3 ? 10 * r : 32 + 10 * (r - 3));
    cnt[n] = x; dep[n] = 0ull; run[n] = 0; if (has_front) front[n] = 0u;
    if (has_heat) heat[n] = 0x8000u;                         // (not hot at tile 0)
  }
  for (int32_t k = wave; k < K; k += KAS_WIDE_WAVES) ring[k * 64 + lane].tag = KAS_TAG_FREE;
  for (int32_t k = wave; k < KAS_WIDE_HOT * (1 + NB); k += KAS_WIDE_WAVES) rank_owner_all[k * 64 + lane] = 0u;
  if (wave == 2 && lane < 2) lstate[lane] = 0u;
  kasw::sync();
  if (wave == 0 && lane == 0) {
    // the padding holder's row: counts that never matter, commits == the dummy ticket
    cnt[nmax] = ((uint64_t)KAS_WIDE_DUMMY_TICKET << KAS_WIDE_COMMIT_SHIFT) | ((uint64_t)0xfffffu << 32) | 0x3fffffffull;
    gdig[0] = 0ull;
    *wd = 0u;
  }
  kasw::sync();
  int32_t wd_idle = 0;

  if (wave == KAS_WIDE_CHAIN_SOLVER || wave >= 3) {
    // ------------------------------------------------------------------ solvers: LDS only
    // a lane without a row claims the next entry of its share of its class's list (rows in row order)
    const int32_t cls = wave == KAS_WIDE_CHAIN_SOLVER ? 1 : 0;
    const int32_t stride = cls ? 1 : NB, first = cls ? 0 : wave - 3;    // my entries: first, first + stride, ...
    uint32_t* rank_owner = rank_owner_all + (cls ? 0 : 1 + first) * (64 * KAS_WIDE_HOT);
    const uint16_t* my_list = clist + cls * (K * 64);
    int32_t e[W], Lp = 0, rot = 0;
#pragma unroll
    for (int q = 0; q < W; ++q) e[q] = dummy_e;
    int32_t cn = 0, my_slot = lane;                          // list entries claimed; slot of the row in hand
    bool cv = false, gfin = false;
    int64_t n_iter = 0, n_blocked = 0, n_relax = 0, n_run_rows = 0, n_runs = 0, n_closure = 0;
    int32_t run_skip = 0, run_backoff = 0;                   // wave-uniform
    uint32_t fstep = 0u;                                     // stamp of the front[] entries of the current step (never 0)
#ifdef KAS_WIDE_DIAG
    int64_t dg_hold = 0, dg_cand = 0, dg_qlen = 0, dg_inhand = 0, dg_c1 = 0, dg_c2 = 0, dg_c4 = 0, dg_steps = 0, dg_tclos = 0, dg_tjac = 0;
#endif
    const int64_t t_begin = kasw::clock_ticks();
    kasw::set_priority<3>();
    for (;;) {
      kasw::repoll();
      n_iter += 1;
      // the counter rows of the row in hand
      int32_t c[W][W];                                      // c[k][r] = count[holder k][replica index r]
      uint32_t d[W];                                        // rows still ahead of mine on holder k
      uint32_t xlo[W], xhi[W];                              // the counter rows as loaded (joint solve: packed adds)
#pragma unroll
      for (int q = 0; q < W; ++q) {
        const uint64_t x = *(const uint64_t*)(lds_raw + (e[q] & 0xffff));
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        const uint32_t f[5] = {lo & KAS_WIDE_FIELD_MASK, (lo >> 10) & KAS_WIDE_FIELD_MASK, (lo >> 20) & KAS_WIDE_FIELD_MASK,
                               hi & KAS_WIDE_FIELD_MASK, (hi >> 10) & KAS_WIDE_FIELD_MASK};
#pragma unroll
        for (int r = 0; r < W; ++r) c[q][r] = (int32_t)f[r];   // (the last index is never compared: dead code)
        d[q] = ((uint32_t)e[q] >> 16) - (hi >> (KAS_WIDE_COMMIT_SHIFT - 32));   // ticket - commits on the node
        xlo[q] = lo; xhi[q] = hi;
      }
      uint32_t d_any = 0u, d_sum = 0u;
#pragma unroll
      for (int q = 0; q < W; ++q) { d_any |= d[q]; d_sum += d[q]; }
      bool ready = cv && d_any == 0u;
      bool ready_q = false;                                 // decided inside a queue in this step
      // ---- joint solve.  Up to KAS_WIDE_HOT nodes are named by rows that wait on one node only (the
      // brokers first fit is filling).  A row in hand is ELIGIBLE when it waits on named nodes only and,
      // on each of them, every row ahead of it (ranks 0 .. its own - 1, rank = ticket - commits) is in
      // this wave's hand and eligible too; the eligible set is then closed under "earlier uncommitted row
      // on one of my nodes", so deciding it in row order is all the sequential algorithm would do next
      // on these nodes.  The counts row i sees on a named node are the node's counts now plus the
      // wins of the eligible rows ahead of it there: picks -> per-position prefix sums over each node's
      // rank layout -> picks are re-evaluated until nothing changes (the earliest row is right after
      // round 1, a row of dependency depth k after round k), and the whole set commits in this step —
      // rows that sit in two queues included.
      bool have_pos = false;                                // wave-uniform: pos[] is final for every row that commits
      int32_t pos[W];
#pragma unroll
      for (int r = 0; r < W; ++r) pos[r] = 0;
      {
        uint64_t nb = 0ull;
        if (run_skip > 0) run_skip -= 1;
        else nb = kasw::ballot(cv && d_any > (uint32_t)KAS_WIDE_VOTE_DEPTH);   // a row deep in some queue
        if (nb != 0ull) {
          constexpr int KH = KAS_WIDE_HOT;
#ifdef KAS_WIDE_DIAG
          const int64_t dg_t0 = kasw::clock_ticks();
#endif
          int32_t my_ax = 0;
          {                                                 // a row names the node on which most rows are ahead of it
            uint32_t dm = 0u;
#pragma unroll
            for (int q = 0; q < W; ++q) { my_ax = d[q] > dm ? (e[q] & 0xffff) : my_ax; dm = d[q] > dm ? d[q] : dm; }
          }
          int32_t hq[KH];                                   // where named node h sits in my (ascending) list, or -1
          uint32_t kx[KH], d_hot = 0u;                      // my rank in its queue
          {
            uint64_t rest = nb;
            // the KH most named nodes among the first KH + KAS_WIDE_VOTE candidates (names of the lowest lanes)
            constexpr int KC = KH + KAS_WIDE_VOTE;
            int32_t cand_ax[KC], cand_n[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k) {
              cand_ax[k] = -1 - k; cand_n[k] = 0;
              if (rest != 0ull) {
                cand_ax[k] = kasw::read_lane(my_ax, kasw::first_lane(rest));
                const uint64_t sup = kasw::ballot(my_ax == cand_ax[k]) & nb;
                cand_n[k] = kasw::popc(sup);
                rest &= ~sup;
              }
            }
#pragma unroll
            for (int h = 0; h < KH; ++h) {
              int32_t ax = -1;                              // (no holder entry has this address)
              {
                int32_t best = 0;
#pragma unroll
                for (int k = 0; k < KC; ++k) { ax = cand_n[k] > best ? cand_ax[k] : ax; best = cand_n[k] > best ? cand_n[k] : best; }
#pragma unroll
                for (int k = 0; k < KC; ++k) cand_n[k] = cand_ax[k] == ax ? 0 : cand_n[k];
              }
              hq[h] = -1;
              uint32_t kk = 0u;                             // (found together with the position: one chain of selects)
#pragma unroll
              for (int q = 0; q < W; ++q) {
                const bool hit = (e[q] & 0xffff) == ax;
                hq[h] = hit ? q : hq[h];
                kk = hit ? d[q] : kk;
              }
              kx[h] = kk;
              d_hot += kx[h];
            }
          }
          // Side dependency (class-1 solver): besides named nodes a row may wait on ONE other node with
          // exactly one row ahead of it there, if that row is in this wave's hand too (two rows of the
          // hand that share an old broker: at configs[4] one row in five of a full hand).  The row ahead
          // is next to commit on that node, so it sees the node's counts as they are and the row behind
          // sees them plus its one increment.
          bool side = false, one = false;
          int32_t qs = 0, pl = lane, pk = 0;                // my list position of that node; lane and list position of the row ahead
          const bool use_side = KAS_WIDE_SIDE && cls != 0 && has_front;   // wave-uniform
          if (use_side) {
            fstep = (fstep + 1u) & 0x1fffffu;
            if (fstep == 0u) {                              // the stamp wrapped: forget every old entry
              for (int32_t n = lane; n <= nmax; n += 64) front[n] = 0u;
              fstep = 1u;
              kasw::lockstep();
            }
#pragma unroll
            for (int q = 0; q < W; ++q)
              if (cv && q < Lp && d[q] == 0u) front[(e[q] & 0xffff) >> 3] = (fstep << 11) | ((uint32_t)q << 8) | (uint32_t)lane;
            one = cv && d_sum == d_hot + 1u;                // one row ahead on one node that is not named
#pragma unroll
            for (int q = 0; q < W; ++q) {
              bool named = false;
#pragma unroll
              for (int h = 0; h < KH; ++h) named = named || hq[h] == q;
              qs = (one && d[q] == 1u && !named) ? q : qs;
            }
          }
          // every row that may be eligible enters the rank layout of its named nodes (one LDS round trip
          // together with the front[] words); rows drop out of the set below, the layout stays
          bool elig = cv && (d_sum == d_hot || one);
#pragma unroll
          for (int h = 0; h < KH; ++h) elig = elig && kx[h] < 64u;
          n_runs += 1;
          const uint32_t seq = (uint32_t)(n_runs & 0xffffff);               // never 0: stale and initial entries differ
#pragma unroll
          for (int h = 0; h < KH; ++h)
            if (elig && hq[h] >= 0) rank_owner[h * 64 + (int32_t)kx[h]] = (seq << 8) | (uint32_t)lane;
          kasw::lockstep();
          uint32_t ow[KH];
          bool have[KH];
          int32_t qlen[KH], own[KH];
#pragma unroll
          for (int h = 0; h < KH; ++h) ow[h] = rank_owner[h * 64 + lane];   // rank view: lane = rank
          if (use_side) {
            const uint32_t f = one ? front[(sel<W>(e, qs) & 0xffff) >> 3] : 0u;
            side = one && (f >> 11) == fstep;
            pl = side ? (int32_t)(f & 0xffu) : lane;
            pk = (int32_t)((f >> 8) & 7u);
            elig = elig && (side || !one);                  // waits on named nodes (and its side dependency) only
          }
          kasw::lockstep();
#pragma unroll
          for (int h = 0; h < KH; ++h) {
            have[h] = (ow[h] >> 8) == seq;
            own[h] = have[h] ? (int32_t)(ow[h] & 0xffu) : lane;
          }
#ifdef KAS_WIDE_DIAG
          const bool elig_first = elig;
          {
            const bool one_d = cv && d_sum == d_hot + 1u;
            dg_c1 += kasw::popc(kasw::ballot(cv && d_sum > d_hot + 1u));          // more than one row ahead on nodes that are not named
            dg_c2 += kasw::popc(kasw::ballot(one_d && !side));                    // one row ahead, not in my hand
          }
#endif
          for (;;) {                                        // closure: drop rows behind a gap in one of their queues
            n_closure += 1;
            const int32_t mine = elig ? 1 : 0;
            int32_t ev[KH], ahead = 1;
#pragma unroll
            for (int h = 0; h < KH; ++h) ev[h] = kasw::shfl(mine, own[h]);   // is the row of rank `lane` still in the set
            if (KAS_WIDE_SIDE) ahead = kasw::shfl(mine, pl);   // ... and the row ahead on my side node (issued with the shuffles
                                                               // above: no branch of its own; pl = my lane without a side table)
            bool ok = elig && (ahead != 0 || !side);
#pragma unroll
            for (int h = 0; h < KH; ++h) {
              const uint64_t hb = kasw::ballot(have[h] && ev[h] != 0);
              qlen[h] = ~hb != 0ull ? kasw::first_lane(~hb) : 64;           // ranks 0..qlen-1 are all eligible rows in hand
              ok = ok && (hq[h] < 0 || (int32_t)kx[h] < qlen[h]);
            }
            if (kasw::ballot(elig && !ok) == 0ull) break;
            elig = ok;
          }
          const bool extra = elig && d_any != 0u;           // decided here, not ready by itself
          const int32_t gain = kasw::popc(kasw::ballot(extra));
#ifdef KAS_WIDE_DIAG
          dg_inhand += kasw::popc(kasw::ballot(cv));
          dg_hold += kasw::popc(kasw::ballot(cv && (hq[0] >= 0 || hq[KH - 1] >= 0)));
          dg_cand += kasw::popc(kasw::ballot(cv && d_sum == d_hot));
          dg_qlen += kasw::popc(kasw::ballot(elig));
          dg_c4 += kasw::popc(kasw::ballot(elig_first && !elig));                 // behind a gap in a queue
          dg_steps += 1;
#endif
          if (gain < KAS_WIDE_MIN_GAIN) {
            run_backoff = run_backoff == 0 ? 1 : (run_backoff < KAS_WIDE_BACKOFF_MAX ? 2 * run_backoff : KAS_WIDE_BACKOFF_MAX);
            run_skip = KAS_WIDE_BACKOFF_MAX > 0 ? run_backoff : 0;
          } else {
            if (gain > KAS_WIDE_MIN_GAIN) run_backoff = 0;
            run_skip = KAS_WIDE_SKIP_AFTER_PASS;
#ifdef KAS_WIDE_DIAG
            const int64_t dg_t1 = kasw::clock_ticks();
            dg_tclos += dg_t1 - dg_t0;
#endif
            // wins ahead of me on named node h, per replica index, packed like the low word of a counter
            // row: index 0 at bit 0, 1 at bit 10, 2 at bit 20 (6 bits each: < 64 rows in a queue), 3 at
            // bit 26.  A virtual count never leaves its 10-bit field: count + rows ahead of me on the node
            // <= rows the node holds at the end < 1023.
            uint32_t pp[KH];
            bool on[KH], mq[KH][W];
            int32_t sh3[KH], src_in[KH], src_out[KH];
#pragma unroll
            for (int h = 0; h < KH; ++h) {
              pp[h] = 0u;
              on[h] = elig && hq[h] >= 0;
              sh3[h] = 3 * (hq[h] >= 0 ? hq[h] : 0);
              src_in[h] = own[h];                                          // owner -> rank view
              src_out[h] = on[h] ? (int32_t)kx[h] : lane;                   // rank view -> owner
#pragma unroll
              for (int q = 0; q < W; ++q) mq[h][q] = hq[h] == q;
            }
            bool sq[W];
#pragma unroll
            for (int q = 0; q < W; ++q) sq[q] = side && qs == q;
            int32_t rs = 7;                                 // the replica index my side node takes in the row ahead
            for (;;) {
              n_relax += 1;
              const uint32_t s_lo = rs < 3 ? (1u << (10 * rs)) : 0u;   // (7 = no side dependency)
              const uint32_t s_hi = rs == 3 ? 1u : 0u;
              int32_t c2[W][W];
#pragma unroll
              for (int q = 0; q < W; ++q) {
                uint32_t lo2 = xlo[q] + (sq[q] ? s_lo : 0u), hi2 = xhi[q] + (sq[q] ? s_hi : 0u);
#pragma unroll
                for (int h = 0; h < KH; ++h) {
                  lo2 += mq[h][q] ? (pp[h] & 0x03ffffffu) : 0u;
                  hi2 += mq[h][q] ? (pp[h] >> 26) : 0u;
                }
                c2[q][0] = (int32_t)(lo2 & KAS_WIDE_FIELD_MASK);
                c2[q][1] = (int32_t)((lo2 >> 10) & KAS_WIDE_FIELD_MASK);
                c2[q][2] = (int32_t)((lo2 >> 20) & KAS_WIDE_FIELD_MASK);
                c2[q][3] = (int32_t)(hi2 & KAS_WIDE_FIELD_MASK);
                if (W > 4) c2[q][W - 1] = 0;                // (the last index is never compared)
              }
              pick_row_packed<W>(c2, Lp, elig, rot, pos);
              int32_t inv = 0;                              // replica index of each list position, 3 bits each
#pragma unroll
              for (int r = 0; r < W; ++r) inv |= (r < Lp ? r : 0) << (3 * pos[r]);   // (every holder gets one pick)
              bool moved = false;
              // (the side shuffle is issued with the ones below, not behind a branch of its own: a branch is a basic
              // block, and the compiler waits for the LDS at its end — one exposed round trip per relaxation round;
              // without a side table pl is the lane itself and side is false)
              const int32_t got = KAS_WIDE_SIDE ? kasw::shfl(inv, pl) : 0;
              int32_t vin[KH];
#pragma unroll
              for (int h = 0; h < KH; ++h) {
                const int32_t rh = on[h] ? ((inv >> sh3[h]) & 7) : 7;     // the replica index node h takes in my row
                vin[h] = kasw::shfl(rh, src_in[h]);
              }
              if (KAS_WIDE_SIDE) {
                const int32_t nrs = side ? ((got >> (3 * pk)) & 7) : 7;
                moved = nrs != rs;
                rs = nrs;
              }
#pragma unroll
              for (int h = 0; h < KH; ++h) {
                const int32_t v = vin[h];
                const int32_t v2 = lane < qlen[h] ? v : 7;
                uint32_t pack = (uint32_t)kasw::count_below(kasw::ballot(v2 == 0));
                pack |= (uint32_t)kasw::count_below(kasw::ballot(v2 == 1)) << 10;
                pack |= (uint32_t)kasw::count_below(kasw::ballot(v2 == 2)) << 20;
                pack |= (uint32_t)kasw::count_below(kasw::ballot(v2 == 3)) << 26;
                const uint32_t back = (uint32_t)kasw::shfl((int32_t)pack, src_out[h]);
                const uint32_t npp = on[h] ? back : 0u;
                moved = moved || npp != pp[h];
                pp[h] = npp;
              }
              if (kasw::ballot(moved) == 0ull) break;
            }
#ifdef KAS_WIDE_DIAG
            dg_tjac += kasw::clock_ticks() - dg_t1;
#endif
            n_run_rows += extra ? 1 : 0;
            ready_q = elig;
            have_pos = true;
          }
        }
      }
      ready = ready || ready_q;
      if (ready) {                                          // (a step in which nothing is ready skips all of it)
        if (!have_pos) pick_row_packed<W>(c, Lp, cv, rot, pos);          // (wave-uniform test)
        // updateCountersFromList (KAS:254-261): count[node][r] += 1 (positions behind the list: + 0)
        int32_t tag = KAS_WTAG_DONE | (Lp << 15);
#pragma unroll
        for (int r = 0; r < W; ++r) {
          const int32_t ad = sel<W>(e, pos[r]) & 0xffff;
          kasw::lds_atomic_add_u64((uint64_t*)(lds_raw + ad), r < Lp ? wide_field_unit(r) : 0ull);
          tag |= (r < Lp ? pos[r] : 0) << (3 * r);
        }
        ring[my_slot].tag = tag;
        cv = false;
      }
      {
        const bool need = !cv && !gfin;
        const uint64_t nbm = kasw::ballot(need);
        const int32_t k = kasw::count_below(nbm);
        const int32_t at = (cn + k) * stride + first;
        const uint32_t st_raw = kasw::load_shared_u32(&lstate[cls]);
        kasw::repoll();                                      // list and slots are read after the count that covers them
        // (the list entry is read right behind the count, not behind the test on it: the LDS serves a wave's reads in
        // issue order, so an entry the count covers is the one the staging wave wrote before it published the count —
        // and the two reads are one round trip instead of two on the step's critical path)
        const int32_t listed_slot = kasw::pinned((int32_t)my_list[at & (K * 64 - 1)]);
        const uint32_t st = (uint32_t)kasw::pinned((int32_t)st_raw);   // (first use of the count: behind the second read's issue)
        const int32_t staged = (int32_t)(st & 0x7fffffffu);
        const bool take = need && at < staged;
        const int32_t slot = take ? listed_slot : my_slot;
        const WideSlot sl = ring[slot];
        if (take) {
#pragma unroll
          for (int q = 0; q < W; ++q) e[q] = sl.e[q];
          Lp = (sl.tag >> 26) & 7;
          rot = sl.rot;
          my_slot = slot;
          cv = true;
        }
        cn += kasw::popc(kasw::ballot(take));
        gfin = (st >> 31) != 0u && cn * stride + first >= staged;   // the list is complete and my share claimed
      }
      const bool fin = gfin && !cv;
      if (kasw::ballot(!fin) == 0) break;
      const bool progress = kasw::ballot(ready) != 0;
      if (watchdog_poll<KAS_WIDE_SPIN_BOUND>(wd, progress, wd_idle)) break;
      if (!progress) { n_blocked += 1; kasw::spin_pause(); }
    }
    if (a.stats && have_s) {
      const int32_t run_rows = kasw::wave_sum((int32_t)n_run_rows);
      if (lane == 0) {
        int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
        if (cls == 1) {
          st[8] = kasw::clock_ticks() - t_begin; st[9] = n_iter; st[10] = n_relax; st[11] = n_blocked;
          st[14] = run_rows; st[6] = n_closure;
#ifdef KAS_WIDE_DIAG
          st[4] = dg_hold; st[5] = dg_cand; st[7] = dg_qlen; st[3] = dg_inhand;
          st[0] = dg_c1; st[1] = dg_c2; st[2] = dg_c4; st[13] = dg_steps;
#ifdef KAS_WIDE_DIAG_TIMES
          st[0] = dg_tclos; st[1] = dg_tjac;                 // ticks before / inside the relaxation loop of the joint solve
#endif
#endif
        } else if (first == 0) {
          st[15] = n_iter;                                   // steps of the first class-0 solver
        }
      }
    }
  } else if (wave == KAS_WIDE_STAGER) {
    // ------------------------------------------------------------------ stager: tickets + staging
    const uint64_t mybit = 1ull << lane;
    const uint64_t lt = mybit - 1ull;
    const uint32_t pad = (uint32_t)nmax;
    TileIter itl = tile_iter_begin(have_s);
    int32_t jl = 0;
    bool endl = false, pf_valid = false, pf_end = false;
    MidRaw<W> pf_raw;                                       // the read-ahead tile's mid row as loaded
#pragma unroll
    for (int q = 0; q < W; ++q) pf_raw.w[q] = ~0u;
    int32_t pf_rot = 0, pf_ow = 0;
    int32_t listed[2] = {0, 0};                             // rows appended to the two claim lists (wave-uniform)
    int64_t f_iter = 0, f_idle = 0;
    for (;;) {
      kasw::repoll();
      f_iter += 1;
      const bool slot_free = ring[(jl & (K - 1)) * 64 + lane].tag == KAS_TAG_FREE;
      const bool room = kasw::ballot(slot_free) == ~0ull;
      const bool stalled = KAS_TEST_STALL_AFTER > 0 && jl >= KAS_TEST_STALL_AFTER;   // debug-build test hook
      const bool staging = !endl && room && pf_valid && !stalled;
      const bool staging_end = staging && pf_end;
      uint32_t cs[W];
      {
        int32_t cells[W];
        mid_unpack<W>(pf_raw, pf_ow, cells);                // -1 = none: sorts last
#pragma unroll
        for (int q = 0; q < W; ++q) cs[q] = staging ? (uint32_t)cells[q] : ~0u;
      }
      const int32_t rotw = pf_rot;
      endl = endl || staging_end;
      pf_valid = pf_valid && !staging;
      if (!pf_valid && !endl) {                             // read ahead: the tile after that
        pf_valid = true;
#pragma unroll
        for (int q = 0; q < W; ++q) pf_raw.w[q] = ~0u;
        pf_ow = 0;
        if (tile_next<64, true>(itl, a, sd)) {
          const int32_t p = itl.row0 + lane;
          pf_rot = itl.rot; pf_ow = itl.tow;
          pf_raw = mid_load_raw<W>((const uint16_t*)a.out + tile_mid_offset(itl), itl.tow, p < itl.tP ? p : 0, p < itl.tP);
        } else {
          pf_end = true;
        }
      }
      if (kasw::ballot(staging) == 0) {                     // nothing to stage: skip the ticket work, poll again
        if (kasw::ballot(!endl) == 0) break;
        if (watchdog_poll<KAS_WIDE_SPIN_BOUND>(wd, false, wd_idle)) break;
        f_idle += 1;
        kasw::nap<KAS_IDLE_NAP>();
        continue;
      }
      // ---- holders ascending (Sets.newTreeSet, KAS:228); empty cells sort last
#pragma unroll
      for (int pass = 0; pass < W; ++pass) {
#pragma unroll
        for (int k = (pass & 1); k + 1 < W; k += 2) {
          const uint32_t lo = cs[k] < cs[k + 1] ? cs[k] : cs[k + 1];
          const uint32_t hi = cs[k] < cs[k + 1] ? cs[k + 1] : cs[k];
          cs[k] = lo; cs[k + 1] = hi;
        }
      }
      int32_t Lp = 0;
      uint32_t hn[W];
#pragma unroll
      for (int q = 0; q < W; ++q) { Lp += cs[q] < pad ? 1 : 0; hn[q] = cs[q] < pad ? cs[q] : pad; }
      // ---- tickets for the tile (wave-wide lockstep): one 64-bit lane mask per node
#pragma unroll
      for (int q = 0; q < W; ++q) kasw::lds_atomic_or_u64(&dep[hn[q]], mybit);
      kasw::lockstep();
      uint64_t m[W];
      uint32_t base[W];
#pragma unroll
      for (int q = 0; q < W; ++q) { m[q] = dep[hn[q]]; base[q] = (uint32_t)run[hn[q]]; }
      kasw::lockstep();
      uint32_t tk[W];
      bool dense = false;                                   // a node of mine that many rows of this tile hold
      uint32_t ht[W];
#pragma unroll
      for (int q = 0; q < W; ++q) ht[q] = has_heat ? (uint32_t)heat[hn[q]] : 0x8000u;
#pragma unroll
      for (int q = 0; q < W; ++q) {
        const bool dq = q < Lp && kasw::popc(m[q]) >= KAS_WIDE_CHAIN_DENSITY;
        dense = dense || dq || (q < Lp && has_heat && ((uint32_t)(jl - (int32_t)ht[q]) & 0xffffu) <= (uint32_t)KAS_WIDE_HEAT);
        tk[q] = base[q] + (uint32_t)kasw::count_below(m[q]);
        // the lowest lane holding the node moves its running count on and clears the mask
        const uint32_t wn = (m[q] & lt) == 0ull ? hn[q] : pad;
        run[wn] = (uint16_t)(base[q] + (uint32_t)kasw::popc(m[q]));
        dep[wn] = 0ull;
        if (has_heat && dq) heat[wn] = (uint16_t)jl;
      }
      kasw::lockstep();                                     // the next tile's masks start from zero
      if (staging) {
        WideSlot o;
#pragma unroll
        for (int q = 0; q < 5; ++q)
          o.e[q] = q < W ? (int32_t)(((q < Lp ? tk[q < W ? q : 0] : (uint32_t)KAS_WIDE_DUMMY_TICKET) << 16) |
                                     (hn[q < W ? q : 0] * 8u))
                         : dummy_e;
        o.rot = rotw; o.spare = 0;
        o.tag = staging_end ? KAS_TAG_END : ((jl & KAS_TAG_JMASK) | (Lp << 26));
        ring[(jl & (K - 1)) * 64 + lane] = o;
      }
      {
        // claim lists: the slot of every staged row goes to the end of its class's list, then the
        // list's length is published (this wave's LDS writes land in issue order: slot, entry, length)
        const bool row = staging && !staging_end;
        const uint64_t b1 = kasw::ballot(row && dense), b0 = kasw::ballot(row && !dense);
        kasw::lockstep();
        if (row) {
          const int32_t at = dense ? listed[1] + kasw::count_below(b1) : listed[0] + kasw::count_below(b0);
          clist[(dense ? K * 64 : 0) + (at & (K * 64 - 1))] = (uint16_t)((jl & (K - 1)) * 64 + lane);
        }
        listed[0] += kasw::popc(b0); listed[1] += kasw::popc(b1);
        kasw::lockstep();
        const uint32_t endbit = kasw::ballot(staging_end) != 0ull ? 0x80000000u : 0u;
        if (lane < 2) kasw::store_shared_u32(&lstate[lane], (uint32_t)(lane == 0 ? listed[0] : listed[1]) | endbit);
        if (staging) jl += 1;
      }
      if (kasw::ballot(!endl) == 0) break;
      if (watchdog_poll<KAS_WIDE_SPIN_BOUND>(wd, true, wd_idle)) break;
    }
    if (a.stats && have_s && lane == 0) {
      int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
      st[12] = f_iter; st[13] = f_idle;
    }
  } else {
    // ------------------------------------------------------------------ retirer: finished rows ->
    // broker ids, digest, the final out row (two batches of rows per lane alternate so that the id
    // reads of one are in flight while the other is written)
    struct Retired { bool on; int32_t id[W], Lp, p, k, ow; int32_t* row; };
    TileIter itr = tile_iter_begin(have_s);
    int32_t jr = 0;
    bool fin = false;
    uint64_t digest = 0;
    auto finish = [&](Retired& r) {
      if (r.on) {
#pragma unroll
        for (int q = 0; q < W; ++q) {
          if (q < r.Lp) {
            r.row[q] = r.id[q];
            digest += kas_digest_cell((uint32_t)r.k, (uint32_t)r.p, (uint32_t)q, r.id[q]);
          }
        }
        if (r.Lp < r.ow) {                                  // -1 behind a list shorter than the row (rare)
#pragma unroll
          for (int q = 0; q < W; ++q) if (q >= r.Lp && q < r.ow) r.row[q] = -1;
        }
      }
      r.on = false;
    };
    auto gather = [&](Retired& r) -> bool {
      if (fin) return false;
      const WideSlot sl = ring[(jr & (K - 1)) * 64 + lane];
      if (KAS_WTAG_IS_DONE(sl.tag)) {
        ring[(jr & (K - 1)) * 64 + lane].tag = KAS_TAG_FREE;
        jr += 1;
        tile_next<64, true>(itr, a, sd);
        int32_t es[W];
#pragma unroll
        for (int q = 0; q < W; ++q) es[q] = sl.e[q];
        r.on = true; r.Lp = (sl.tag >> 15) & 7; r.p = itr.row0 + lane; r.k = itr.k;
        r.ow = r.p < itr.tP ? itr.tow : 0;                  // a lane past the last row of the tile writes nothing
        r.row = a.out + itr.tout + (int64_t)r.p * itr.tow;
#pragma unroll
        for (int q = 0; q < W; ++q) {
          const int32_t en = sel<W>(es, (sl.tag >> (3 * q)) & 7);
          const int32_t node = q < r.Lp ? (en & 0xffff) >> 3 : 0;
          r.id[q] = g_node_id[node];                        // node table of the scenario: L2-resident
        }
        return true;
      }
      if (sl.tag == KAS_TAG_END) fin = true;
      return false;
    };
    Retired ra, rb;
    ra.on = false; rb.on = false; ra.Lp = 0; rb.Lp = 0; ra.p = 0; rb.p = 0; ra.k = 0; rb.k = 0; ra.ow = 0; rb.ow = 0;
    ra.row = nullptr; rb.row = nullptr;
#pragma unroll
    for (int q = 0; q < W; ++q) { ra.id[q] = 0; rb.id[q] = 0; }
    for (;;) {
      bool retired = false;
      kasw::repoll();
      finish(ra);
      retired = gather(ra) || retired;
      kasw::repoll();
      finish(rb);
      retired = gather(rb) || retired;
      if (kasw::ballot(!fin) == 0) break;
      const bool progress = kasw::ballot(retired) != 0;
      if (watchdog_poll<KAS_WIDE_SPIN_BOUND>(wd, progress, wd_idle)) break;
      if (!progress) kasw::nap<KAS_IDLE_NAP>();
    }
    finish(ra); finish(rb);
    kasw::lds_atomic_add_u64(&gdig[0], digest);
    kasw::lockstep();
    if (have_s && lane == 0) a.scenario_results[s].digest = gdig[0];
    // the Context goes back: every row has been retired, so every commit is in the counter rows
    for (int32_t n = lane; n < sd.n_nodes && ccols > 0; n += 64) {
      const uint64_t x = cnt[n];
      for (int32_t r = 0; r < ccols; ++r)
        g_ctx[(int64_t)n * sd.ctx_width + r] = (int32_t)((x >> (r < 3 ? 10 * r : 32 + 10 * (r - 3))) & KAS_WIDE_FIELD_MASK);
    }
    // ... and without a Context, where the plan could not bound the counts a priori: did every count stay
    // inside the 10 bits it is read with?  (header of this file)
    if ((a.flags & KAS_FLAG_WIDE_CHECK) && have_s && ccols == 0 && a.ord_flag) {
      bool bad = false;
      for (int32_t n = lane; n < sd.n_nodes; n += 64) {
        const uint64_t x = cnt[n];
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        const uint32_t f2 = lo >> 20, f4 = (hi >> 10) & 0x7ffu;                  // (with their room)
        const uint32_t sum = (lo & KAS_WIDE_FIELD_MASK) + ((lo >> 10) & KAS_WIDE_FIELD_MASK) + f2 +
                             (hi & KAS_WIDE_FIELD_MASK) + f4;
        bad = bad || sum != (hi >> (KAS_WIDE_COMMIT_SHIFT - 32)) || f2 > KAS_WIDE_FIELD_MASK ||
              (W > 4 ? false : f4 != 0u);
      }
      if (kasw::ballot(bad) != 0ull && lane == 0) a.ord_flag[s] = 1;
    }
    if (KAS_WIDE_SPIN_BOUND > 0 && have_s && lane == 0 && *(volatile uint32_t*)wd != 0u) {
      a.scenario_results[s].status = KAS_FAIL_WATCHDOG;
      a.scenario_results[s].fail_topic = -1; a.scenario_results[s].fail_partition = -1;
    }
  }
}

}  // namespace kas
