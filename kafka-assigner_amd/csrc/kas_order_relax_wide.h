// kas_order_relax_wide.h — order kernel, RELAXATION form for lists 4 and 5 wide: P5 (computePreferenceLists, KAS:202-239 with the
// PreferenceListOrderTracker of KAS:244-302), one wavefront per scenario.  Included at the end of kas_solver_body.h.
//
// The form is kas_order_relax.h's (read its header first): 64 consecutive rows are a tile, lane i evaluates row i, and inside
// the tile the sequential answer is the fixed point of
//     outcome(i) = picks( committed counts + what the rows j < i of the tile that hold the same node add ),
// where "what the earlier rows add" comes out of the LDS itself — an atomic add WITH RETURN hands every lane the word before
// the instruction plus the addends of the lower lanes naming the same word (kas_wave.h, lds_add_rtn_u64), and the tile's
// (row, cell) pairs are added in row-major order (pair v = W row + cell is lane v mod 64 of instruction v div 64: W
// instructions a tile) so that lane order is row order.  What is different at these widths:
//
//   counter word   uint64 per node: count[n][0..3] in four 16-bit fields.  count[n][r] is READ only where a pick has two or
//                  more candidates (KAS:263-278), i.e. for r <= L - 2 <= 3 at lists up to 5 wide; every list position r <= 3
//                  ADDS to its field whatever the list's length (KAS:236: a topic 4 wide and a topic 5 wide of one scenario
//                  share the counters), position 4 adds nothing anybody reads.
//   cells          a row's holders are SORTED (ascending node index == ascending broker id, KAS:228's TreeSet) by its row lane
//                  before they go to the pair lanes: cell i of a row is its i-th smallest holder, so the visit order of
//                  KAS:188-200 is arithmetic on i — with A the holders not picked yet, m = |A|, the holder of alive rank a is
//                  visited at position (a + abs(hash) mod m) mod m, and a pick is the minimum of  count << 6 | visit position
//                  << 3 | i  over A ("first strictly smaller in visit order").  No tag tables: the general evaluation is the
//                  only one (rows with fewer holders than the batch's width take it with L < W).
//   row word       3 bits per cell: the list position the cell was picked for (7: none); a pair lane turns its cell's code c
//                  into the addend 1 << 16 c (c <= 3) or 0.
//
// Per tile: sort (9 compare-exchanges at W = 5), then per evaluation W picks of ~12 vector instructions per candidate, W pair
// instructions each way (ds_sub_u64 of the previous addends, ds_add_rtn_u64 of the new ones) and W 8-byte staging words per
// lane.  That is 4-5 times the work of the 3-wide form per evaluation, which is why DESIGN.md section 4.3 had declined to
// build it on an estimate; VERDICT r5 asked for the measurement (profiles/r06_*config5*).
//
// Applicable (KasShape::relaxw_ok): lists 4 or 5 wide, no Context handed in, no topic hash of Integer.MIN_VALUE, fewer than
// 65,535 rows per node, the broker ids fit the LDS beside the counter words (kas_order_relaxw_lds).  Launched where the plan
// says (KAS_PLAN_RELAX_TILES(1) at these widths / KAS_RELAXW_DEFAULT); the wide ticket form (kas_order_wide.h) otherwise.
#pragma once

namespace kas {

#define KAS_RELAXW_PAD_WORD 0xfff0fff0fff0fff0ull   // counter word of the padding node: only ever gets + 0

// The picks of one row in rank space.  x[i]: counter word of the row's i-th smallest holder (i < L); idxp: abs(hash) mod m for
// m = 1..W in 4-bit fields (bits 4 m ..).  Returns 3 bits per list position r: the rank of the holder picked for it (7: the
// row has no such position).
template <int W>
KAS_DEV uint32_t relaxw_eval(const uint64_t (&x)[W], int32_t L, uint32_t idxp) {
  uint32_t A = (1u << L) - 1u;                               // holders not picked yet (nodeSet, KAS:228-232)
  uint32_t picks = 0u;
#pragma unroll
  for (int r = 0; r < W; ++r) {
    const int32_t m = L - r;                                 // |nodeSet|
    const int32_t idx = (int32_t)((idxp >> (4 * (m > 0 ? m : 0))) & 15u);
    uint32_t best = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const int32_t ar = __builtin_popcount(A & ((1u << i) - 1u));       // alive rank: position in the sorted set
      int32_t vp = ar + idx;                                 // order[(idx + a) mod m] = sorted[a]  (KAS:193-197)
      vp -= vp >= m ? m : 0;
      const uint32_t f = r < 4 ? (uint32_t)(x[i] >> (16 * (r < 4 ? r : 0))) & 0xffffu : 0u;   // count[n][r] (a single candidate: not read)
      const uint32_t key = (f << 6) | ((uint32_t)vp << 3) | (uint32_t)i;
      best = (((A >> i) & 1u) != 0u && key < best) ? key : best;
    }
    const uint32_t ks = best & 7u;
    picks |= (m > 0 ? ks : 7u) << (3 * r);
    A &= m > 0 ? ~(1u << ks) : 0xffffffffu;                  // nodeSet.remove (KAS:232)
  }
  return picks;
}

// ascending holders of a row from its mid-row cells (node indices, -1: none): h[0..L) ascending, the rest nmax (the padding node)
template <int W>
KAS_DEV void relaxw_sort(const int32_t (&cells)[W], int32_t nmax, uint32_t (&h)[W], int32_t& L) {
  int32_t t[W];
  sort_holders<W>(cells, t, L);                              // (0x7fffffff behind the holders)
#pragma unroll
  for (int k = 0; k < W; ++k) h[k] = k < L ? (uint32_t)t[k] : (uint32_t)nmax;
}

template <int W>
KAS_DEV void order_relax_wide(const KasLaunch& a, int32_t s, unsigned char* lds_raw) {
  static_assert(W == 4 || W == 5, "four 16-bit count fields serve lists up to 5 wide");
  if constexpr (KAS_RELAX_PRIO > 0) kasw::set_priority<KAS_RELAX_PRIO>();
  const int lane = kasw::lane();
  const kas_scenario_desc sd = a.scen[s];
  const int32_t N = sd.n_nodes;
  const int32_t nmax = a.n_max > 0 ? a.n_max : 1;
  uint64_t* cnt = (uint64_t*)lds_raw;                                     // [nmax + 1]: + the padding node's word
  uint32_t* rbuf = (uint32_t*)(lds_raw + kas_align16(8 * (int64_t)(nmax + 1)));   // [64] row words of the tile
  uint64_t* stage = (uint64_t*)(rbuf + 64);                               // [64 W] by pair of the tile
  uint32_t* idt = (uint32_t*)(stage + 64 * W);                            // [nmax] the scenario's broker ids
  const int32_t* g_node_id = a.node_id + sd.node_off;
  const int64_t t_begin = kasw::clock_ticks();
  for (int32_t n = lane; n < N; n += 64) { cnt[n] = 0ull; idt[n] = (uint32_t)g_node_id[n]; }
  if (lane == 0) cnt[nmax] = KAS_RELAXW_PAD_WORD;
  // my pairs: pair v = 64 t + lane is cell v mod W of row v div W
  const uint32_t* prow[W];
  uint32_t pcell[W];
#pragma unroll
  for (int t = 0; t < W; ++t) {
    const int32_t v = 64 * t + lane, r = v / W;
    prow[t] = rbuf + r;
    pcell[t] = (uint32_t)(3 * (v - W * r));                  // (shift of the cell's code in the row word)
  }
  uint64_t* const pslot = stage + lane;                      // pair t's staging word: pslot[64 t]
  uint64_t* const mine = stage + W * lane;                   // (row lane) my row's W words
  kasw::lockstep();

  uint64_t digest = 0;
  int32_t n_tiles = 0, n_evals = 0;                          // (wave-uniform)
  bool stuck = false, unsound = false;
  uint32_t rowsf[4] = {0u, 0u, 0u, 0u};                      // rows that added to field r (wave-uniform): conservation, below
  for (int32_t k = 0; k < sd.topic_count; ++k) {
    const int32_t ti = sd.topic_begin + k;
    if (a.topic_results[ti].status != KAS_OK) continue;     // (a failed or skipped topic emits nothing)
    const kas_topic_desc td = a.topics[ti];
    const int32_t P = td.n_partitions, ow = td.out_width;
    if (P <= 0) continue;
    int32_t* out = a.out + td.out_off;
    const uint16_t* mid = mid_base(out, P, ow);
    const int32_t nt = (P + 63) >> 6;
    uint32_t idxp = 0u;                                      // abs(hash) mod m, m = 1..W (KAS:190; never negative here: the plan checks)
#pragma unroll
    for (int m = 1; m <= W; ++m) idxp |= (uint32_t)java_abs_mod(td.name_hash, m) << (4 * m);
    // rows come in as mid rows at the end of the topic's out region and leave as final rows from its start (a final row
    // never reaches a mid row that is still to be read: kas_solver_body.h, "Intermediate rows"); the tile after the current
    // one is asked for before the current one is solved
    MidRaw<W> nxr = mid_load_raw<W>(mid, ow, lane < P ? lane : 0, lane < P);
    for (int32_t tile = 0; tile < nt; ++tile) {
      const int32_t p = (tile << 6) + lane;
      const bool active = p < P;
      int32_t cells[W], L;
      uint32_t h[W];
      mid_unpack<W>(nxr, ow, cells);
      relaxw_sort<W>(cells, nmax, h, L);
      {
        const int32_t pn = p + 64;
        nxr = mid_load_raw<W>(mid, ow, pn < P ? pn : 0, pn < P);
      }
      n_tiles += 1;
      // ---- hand the sorted holders to the pair lanes; counter words of my holders as the previous tile left them
      kasw::lockstep();                                      // (the previous tile's words have been read)
#pragma unroll
      for (int i = 0; i < W; ++i) mine[i] = (uint64_t)h[i];
      kasw::lockstep();
      uint64_t* padr[W];
      uint64_t padd[W];
#pragma unroll
      for (int t = 0; t < W; ++t) { padr[t] = cnt + (uint32_t)pslot[64 * t]; padd[t] = 0ull; }
      uint64_t x[W];
#pragma unroll
      for (int i = 0; i < W; ++i) x[i] = cnt[h[i]];
      kasw::lockstep();
      uint32_t oc_prev = 0xffffffffu;
      for (int32_t it = 0;; ++it) {
        n_evals += 1;
        // (lane i is right after evaluation i + 1, so 65 evaluations always suffice: more means the LDS did not hand the
        // additions out in lane order — give up with a status instead of looping)
        if (it > 66) { stuck = true; break; }
        const uint32_t oc = relaxw_eval<W>(x, L, idxp);
        if (kasw::ballot(oc != oc_prev) == 0ull) break;      // nobody's outcome moved: the words hold the tile's commits
        // the row word: per cell the list position it was picked for (7: none; positions beyond 3 add nothing that is read)
        uint32_t rw = 0x7fffu;
#pragma unroll
        for (int r = 0; r < (W < 4 ? W : 4); ++r) {
          const uint32_t ks = (oc >> (3 * r)) & 7u;
          rw = ks != 7u ? ((rw & ~(7u << (3 * ks))) | ((uint32_t)r << (3 * ks))) : rw;
        }
        rbuf[lane] = rw;
        kasw::lockstep();                                    // the row words are written
        uint64_t nadd[W];
#pragma unroll
        for (int t = 0; t < W; ++t) {
          const uint32_t code = (*prow[t] >> pcell[t]) & 7u;
          nadd[t] = code < 4u ? (1ull << (16 * code)) : 0ull;
        }
        if (it > 0) {                                        // (wave-uniform) take the previous additions back
#pragma unroll
          for (int t = 0; t < W; ++t) kasw::lds_sub_u64(padr[t], padd[t]);
        }
        kasw::lockstep();
        uint64_t got[W];
#pragma unroll
        for (int t = 0; t < W; ++t) {
          got[t] = kasw::lds_add_rtn_u64(padr[t], nadd[t]);
          kasw::lockstep();                                  // (one instruction at a time, lanes in order: the hardware's order)
          padd[t] = nadd[t];
        }
#pragma unroll
        for (int t = 0; t < W; ++t) pslot[64 * t] = got[t];
        kasw::lockstep();
#pragma unroll
        for (int i = 0; i < W; ++i) x[i] = mine[i];
        oc_prev = oc;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) rowsf[r] += (uint32_t)kasw::popc(kasw::ballot(L > r));
      // ---- the final row: holders in pick order, as broker ids
      if (active) {
#pragma unroll
        for (int r = 0; r < W; ++r) {
          if (r < ow) {                                      // (wave-uniform)
            const uint32_t ks = (oc_prev >> (3 * r)) & 7u;
            uint32_t node = h[0];
#pragma unroll
            for (int i = 1; i < W; ++i) node = ks == (uint32_t)i ? h[i] : node;
            const int32_t id = r < L ? (int32_t)idt[node < (uint32_t)nmax ? node : 0u] : -1;
            out[(int64_t)p * ow + r] = id;
            if (r < L) digest += kas_digest_cell((uint32_t)k, (uint32_t)p, (uint32_t)r, id);
          }
        }
      }
    }
  }
  {
    // conservation: every row with a list position r <= 3 added exactly one to some node's field r (and the padding node's
    // word is untouched) — a lost or doubled addition, i.e. an LDS that does not serve lanes in order, cannot go unnoticed
    kasw::lockstep();
    uint32_t f[4] = {0u, 0u, 0u, 0u};
    for (int32_t n = lane; n < N; n += 64) {
      const uint64_t w = cnt[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) f[r] += (uint32_t)(w >> (16 * r)) & 0xffffu;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (!stuck && (uint32_t)kasw::wave_sum((int)f[r]) != rowsf[r]) unsound = true;
    if (!stuck && kasw::ballot(cnt[nmax] != KAS_RELAXW_PAD_WORD) != 0ull) unsound = true;
  }
  const uint64_t dsum = kasw::wave_sum_u64(digest);
  if (lane == 0) {
    a.scenario_results[s].digest = dsum;
    if (stuck || unsound) {
      a.scenario_results[s].status = KAS_FAIL_WATCHDOG;
      a.scenario_results[s].fail_topic = -1; a.scenario_results[s].fail_partition = -1;
    }
    if (a.stats) {
      int64_t* st = a.stats + (int64_t)s * KAS_STATS_PER_SCENARIO;
      st[8] = kasw::clock_ticks() - t_begin; st[9] = n_evals; st[12] = n_tiles; st[13] = 0; st[10] = 0; st[11] = unsound ? 1 : 0;
    }
  }
}

}  // namespace kas
