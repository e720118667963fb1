// kas_solver_body.h — the per-scenario solver, one 64-lane wavefront per scenario.
//
// Computes exactly what KafkaAssignmentStrategy.getRackAwareAssignment computes
// (KafkaAssignmentStrategy.java:40-63, "KAS"), for every topic of a scenario in order, with the
// pre-checks of KafkaTopicAssigner.generateAssignment (KafkaTopicAssigner.java:65-69, "KTA").
// The reference is sequential; each phase below is an order-preserving parallel formulation
// whose result is identical to the sequential one:
//
//  P0  cap = (int)ceil((double)(int)(P*rf)/N)                                     KAS:65-71
//  P2  sticky fill (KAS:101-131): sweep r = 0..cur_width-1 over 64-row tiles in ascending row
//      order, lane = row.  A replica is eligible iff its broker is a node and its rack is not
//      held by an earlier accepted replica of the same row.  Eligible lanes whose node is not
//      yet full bump load[n] with one LDS atomic; only when a node overflows inside a tile are
//      that node's lanes ranked by ballot (lane order == row order) and the first cap-load
//      kept.  The accepted lanes of (sweep, tile) are one 64-bit ballot word in HBM scratch.
//  P3  orphans (KAS:133-160): rf - accepted per row; rows are compacted in ascending order
//      into an LDS ring of 64-orphan windows.
//  P4  first fit (KAS:162-186): the reference walks order[0..] for each orphan in turn.  Cell
//      (orphan i, node position j) depends only on (i, j'<j) and (i'<i, j), so the grid is
//      evaluated position-major: for each non-full node in processing order, the lanes that
//      still need a replica and may use that rack ballot, and the first cap-load[n] of them (in
//      lane == orphan order) take it.  Full nodes never become non-full, so only a compacted
//      list of non-full nodes is walked (the reference spends >99% of its probes on them).
//  P5  preference order (KAS:202-239): row p reads count[n][0..L) of its own nodes, picks, then
//      increments L counters, so rows conflict only when they share a node.  64 ascending rows
//      per tile; each round every pending lane claims its nodes with an LDS atomic max keyed
//      (epoch, 63-lane); a lane that owns all its nodes has no earlier pending row sharing a
//      node and commits; the rest retry.  Lane order == row order, so the result is the
//      sequential one.
//
// Everything cross-lane goes through kas_wave.h; all control flow around those calls is
// wave-uniform.  List positions live in registers through fully unrolled loops (W is a
// template parameter) — no dynamically indexed private arrays.
#pragma once
#include <stdint.h>

#include "kas_abi.h"
#include "kas_plan_math.h"
#include "kas_wave.h"

namespace kas {

struct TopicOutcome {
  int32_t status;
  int32_t fail_partition;
  int32_t moved_replicas;
  int32_t moved_partitions;
  uint64_t digest;   // per-lane partial; summed over the wave by the caller
};

struct LdsView {
  int32_t* cnt;
  uint32_t* owner;
  int32_t* load;
  int16_t* rack;
  int16_t* live;
  int16_t* idmap;
  int32_t* ids;
  int32_t* ring_p;
  int32_t* ring_meta;
  int16_t* ring_rack;   // [W][KAS_RING_CAP]
};

struct NodeMap {
  int32_t n;            // N
  int32_t min_id;
  uint32_t range;       // direct-table extent, 0 = binary search
};

// nodeMap.get(nodeId) (KAS:119): node index of a broker id, or -1.
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
  int32_t v = a[0];
#pragma unroll
  for (int j = 1; j < W; ++j) v = (i == j) ? a[j] : v;
  return v;
}

template <int W>
KAS_DEV void put(int32_t (&a)[W], int32_t i, int32_t v) {
#pragma unroll
  for (int j = 0; j < W; ++j) a[j] = (i == j) ? v : a[j];
}

// ---------------------------------------------------------------------------------------------
// P4 window: up to 64 orphans (lane = orphan, ascending row order), position-major first fit.
// Returns -1, or the lane index of the first orphan that cannot be fully assigned (KAS:183).
// ---------------------------------------------------------------------------------------------
template <int W>
KAS_DEV int32_t p4_window(const LdsView& L, int32_t count, int32_t cap, int32_t live_count,
                          int32_t& head, int32_t* out, int32_t ow, int64_t (&st)[8]) {
  const int lane = kasw::lane();
  const bool mine = lane < count;
  const int32_t p = mine ? L.ring_p[lane] : 0;
  const int32_t meta = mine ? L.ring_meta[lane] : 0;
  int32_t need = meta & 0xff;
  int32_t hc = (meta >> 8) & 0xff;
  int32_t hr[W];
#pragma unroll
  for (int j = 0; j < W; ++j) hr[j] = mine ? (int32_t)L.ring_rack[j * KAS_RING_CAP + lane] : -1;
  kasw::sync();   // ring fully read before the caller shifts it

  int32_t fail_lane = -1;
  int32_t j = head;
  st[4] += 1;
  for (;;) {
    uint64_t pend = kasw::ballot(need > 0);
    if (pend == 0) break;
    st[5] += 1;
    if (j >= live_count) { fail_lane = kasw::first_lane(pend); break; }
    const int32_t n = (int32_t)L.live[j];
    const int32_t slots = cap - L.load[n];
    if (slots > 0) {
      const int32_t rk = (int32_t)L.rack[n];
      bool want = need > 0;
#pragma unroll
      for (int k = 0; k < W; ++k) want = want && !(k < hc && hr[k] == rk);
      const uint64_t w = kasw::ballot(want);
      if (w != 0) {
        const int32_t rank = kasw::popc(w & kasw::lanemask_lt());
        if (want && rank < slots) {                      // accept (KAS:178-181)
          out[(int64_t)p * ow + hc] = n;
          put<W>(hr, hc, rk);
          hc += 1;
          need -= 1;
        }
        const int32_t takers = kasw::popc(w);
        kasw::sync();                                    // every lane has read load[n]
        if (lane == 0) L.load[n] += takers < slots ? takers : slots;
        kasw::sync();
      }
    }
    ++j;
  }
  // drop the leading nodes that are now full from future windows
  while (head < live_count && L.load[(int32_t)L.live[head]] >= cap) ++head;
  return fail_lane;
}

// ---------------------------------------------------------------------------------------------
// One topic == one getRackAwareAssignment call.
// ---------------------------------------------------------------------------------------------
template <int W>
KAS_DEV TopicOutcome solve_topic(const KasLaunch& a, const kas_topic_desc& td, uint32_t topic_k,
                                 const LdsView& L, const NodeMap& nm, const int32_t* g_node_id,
                                 const int32_t* g_node_rack, uint64_t* accmask, int64_t (&st)[8]) {
  const int lane = kasw::lane();
  const uint64_t lt = kasw::lanemask_lt();
  const int32_t N = nm.n;
  const int32_t P = td.n_partitions;
  const int32_t cw = td.cur_width;
  const int32_t rf = td.rf;
  const int32_t ow = td.out_width;
  const int32_t hash = td.name_hash;
  const int32_t* cur = a.cur + td.cur_off;
  int32_t* out = a.out + td.out_off;
  const int32_t* len_arr = td.cur_len_off >= 0 ? a.aux + td.cur_len_off : nullptr;
  const int32_t* inp_arr = td.in_partitions_off >= 0 ? a.aux + td.in_partitions_off : nullptr;
  const int32_t* pid_arr = td.part_id_off >= 0 ? a.aux + td.part_id_off : nullptr;
  const int32_t nt = (P + 63) >> 6;

  TopicOutcome res;
  res.status = KAS_OK; res.fail_partition = -1;
  res.moved_replicas = 0; res.moved_partitions = 0; res.digest = 0;

  int64_t tmark = kasw::clock_ticks();
  // ---- P0: capacity (KAS:45, 65-71) over |partitions| ---------------------------------------
  int32_t n_in = P;
  if (inp_arr) {
    int32_t c = 0;
    for (int32_t p = lane; p < P; p += 64) c += inp_arr[p] != 0 ? 1 : 0;
    n_in = kasw::wave_sum(c);
  }
  const int32_t cap = max_replicas_per_node(N, n_in, rf);

  // ---- per-topic node state (KAS:46, 73-99): region A of the LDS carve-up -------------------
  for (int32_t i = lane; i < N; i += 64) {
    L.load[i] = 0;
    L.rack[i] = (int16_t)g_node_rack[i];
  }
  if (nm.range != 0u) {
    for (uint32_t i = (uint32_t)lane; i < nm.range; i += 64u) L.idmap[i] = (int16_t)-1;
    kasw::sync();
    for (int32_t i = lane; i < N; i += 64) L.idmap[(uint32_t)g_node_id[i] - (uint32_t)nm.min_id] = (int16_t)i;
  } else {
    for (int32_t i = lane; i < N; i += 64) L.ids[i] = g_node_id[i];
  }
  kasw::sync();

  { const int64_t now = kasw::clock_ticks(); st[0] += now - tmark; tmark = now; }
  // ---- P2: sticky fill (KAS:49, 101-131) ----------------------------------------------------
  for (int32_t r = 0; r < cw; ++r) {
    for (int32_t tile = 0; tile < nt; ++tile) {
      const int32_t p = (tile << 6) + lane;
      const bool active = p < P;
      const int32_t len = active ? (len_arr ? len_arr[p] : cw) : 0;
      const int32_t* row = cur + (int64_t)p * cw;
      int32_t n = -1;
      if (r < len) n = node_lookup(L, nm, row[r]);         // node != null (KAS:119-120)
      bool elig = n >= 0;
      const int32_t rk = elig ? (int32_t)L.rack[n] : -1;
#pragma unroll
      for (int r2 = 0; r2 < W - 1; ++r2) {
        if (r2 < r) {                                       // wave-uniform
          const uint64_t aw = kasw::load_shared_u64(accmask + (int64_t)r2 * nt + tile);
          if (elig && ((aw >> lane) & 1ull)) {
            const int32_t n2 = node_lookup(L, nm, row[r2]);
            if ((int32_t)L.rack[n2] == rk) elig = false;    // rack.canAccept (KAS:346-348)
          }
        }
      }
      const bool took = elig && L.load[n] < cap;            // size() < capacity (KAS:322)
      kasw::lockstep();                                     // every lane saw the pre-tile load
      if (took) kasw::lds_atomic_add(&L.load[n], 1);
      kasw::sync();
      bool accepted = took;
      uint64_t todo = kasw::ballot(took && L.load[n] > cap);
      if (todo != 0) {
        st[7] += 1;
        // some node overflowed inside this tile: keep its first (cap - load_before) lanes
        while (todo != 0) {
          const int leader = kasw::first_lane(todo);
          const int32_t t = kasw::shfl(n, leader);
          const bool same_l = took && n == t;
          const uint64_t same = kasw::ballot(same_l);
          const int32_t before = L.load[t] - kasw::popc(same);
          if (same_l) accepted = before + kasw::popc(same & lt) < cap;
          kasw::sync();                                     // all lanes read load[t]
          if (lane == leader) L.load[t] = cap;
          todo &= ~same;
        }
        kasw::sync();
      }
      const uint64_t accw = kasw::ballot(accepted);
      if (lane == 0) kasw::store_shared_u64(accmask + (int64_t)r * nt + tile, accw);
    }
    kasw::sync();   // this sweep's mask words are visible to the next sweep's loads
  }

  { const int64_t now = kasw::clock_ticks(); st[1] += now - tmark; tmark = now; }
  // ---- KAS:168: getNodeProcessingOrder(topic, all nodes); runs even with zero orphans -------
  const int32_t idxN = java_abs_mod(hash, N);
  if (idxN < 0) { res.status = KAS_FAIL_HASH_INDEX; return res; }
  const int32_t start = (N - idxN) % N;        // order[j] = sorted[(j + start) % N]

  // non-full nodes in processing order (full nodes can never accept again)
  int32_t live_count = 0;
  for (int32_t base = 0; base < N; base += 64) {
    const int32_t j = base + lane;
    int32_t n = j + start; if (n >= N) n -= N;
    const bool is_live = j < N && L.load[n] < cap;
    const uint64_t m = kasw::ballot(is_live);
    if (is_live) L.live[live_count + kasw::popc(m & lt)] = (int16_t)n;
    live_count += kasw::popc(m);
  }
  kasw::sync();

  // ---- P3 + P4: orphans (KAS:52, 133-160) and first fit (KAS:56, 162-186) -------------------
  int32_t ring_count = 0, head = 0;
  int32_t moved_r = 0, moved_p = 0;
  int32_t fail_row = -1;
  for (int32_t tile = 0; tile < nt && fail_row < 0; ++tile) {
    const int32_t p = (tile << 6) + lane;
    const bool active = p < P;
    const int32_t len = active ? (len_arr ? len_arr[p] : cw) : 0;
    const int32_t* row = cur + (int64_t)p * cw;
    int32_t ids[W];
    int32_t hold[W], hrack[W];
    uint32_t accbits = 0;
    int32_t hc = 0;
#pragma unroll
    for (int r = 0; r < W; ++r) {
      ids[r] = -1; hold[r] = -1; hrack[r] = -1;
    }
#pragma unroll
    for (int r = 0; r < W; ++r) {
      if (r < cw) {                                         // wave-uniform
        const uint64_t aw = kasw::load_shared_u64(accmask + (int64_t)r * nt + tile);
        if (r < len) ids[r] = row[r];
        if (active && ((aw >> lane) & 1ull)) {
          accbits |= 1u << r;
          const int32_t n = node_lookup(L, nm, ids[r]);
          put<W>(hold, hc, n);
          put<W>(hrack, hc, (int32_t)L.rack[n]);
          hc += 1;
        }
      }
    }
    if (active) {
#pragma unroll
      for (int k = 0; k < W; ++k) if (k < ow) out[(int64_t)p * ow + k] = hold[k];   // node indices for now
    }
    const bool in_parts = active && (inp_arr ? inp_arr[p] != 0 : true);
    const int32_t need = in_parts ? (rf - hc > 0 ? rf - hc : 0) : 0;   // KAS:151-157
    // a distinct current broker that was not kept => set(new) != set(cur)
    bool dropped = false;
#pragma unroll
    for (int r = 0; r < W; ++r) {
      if (r < len && !((accbits >> r) & 1u)) {
        bool kept_elsewhere = false;
#pragma unroll
        for (int r2 = 0; r2 < W; ++r2)
          if (r2 < len && ((accbits >> r2) & 1u) && ids[r2] == ids[r]) kept_elsewhere = true;
        if (!kept_elsewhere) dropped = true;
      }
    }
    moved_r += need;
    moved_p += (active && (dropped || need > 0)) ? 1 : 0;

    const bool orphan = need > 0;
    const uint64_t om = kasw::ballot(orphan);
    if (orphan) {
      const int32_t slot = ring_count + kasw::popc(om & lt);
      L.ring_p[slot] = p;
      L.ring_meta[slot] = need | (hc << 8);
#pragma unroll
      for (int k = 0; k < W; ++k) L.ring_rack[k * KAS_RING_CAP + slot] = (int16_t)hrack[k];
    }
    ring_count += kasw::popc(om);
    kasw::sync();
    while (ring_count >= 64 && fail_row < 0) {
      const int32_t fl = p4_window<W>(L, 64, cap, live_count, head, out, ow, st);
      if (fl >= 0) { fail_row = L.ring_p[fl]; break; }
      // shift the ring down by one window
      const int32_t rest = ring_count - 64;
      int32_t tp = 0, tm = 0; int32_t tr[W];
      if (lane < rest) {
        tp = L.ring_p[64 + lane]; tm = L.ring_meta[64 + lane];
#pragma unroll
        for (int k = 0; k < W; ++k) tr[k] = L.ring_rack[k * KAS_RING_CAP + 64 + lane];
      }
      kasw::sync();
      if (lane < rest) {
        L.ring_p[lane] = tp; L.ring_meta[lane] = tm;
#pragma unroll
        for (int k = 0; k < W; ++k) L.ring_rack[k * KAS_RING_CAP + lane] = (int16_t)tr[k];
      }
      ring_count = rest;
      kasw::sync();
    }
  }
  if (fail_row < 0 && ring_count > 0) {
    const int32_t fl = p4_window<W>(L, ring_count, cap, live_count, head, out, ow, st);
    if (fl >= 0) fail_row = L.ring_p[fl];
  }
  { const int64_t now = kasw::clock_ticks(); st[2] += now - tmark; tmark = now; }
  if (fail_row >= 0) {                                       // KAS:183-184
    res.status = KAS_FAIL_UNASSIGNABLE;
    res.fail_partition = pid_arr ? pid_arr[fail_row] : fail_row;
    return res;
  }
  kasw::sync();   // P4's out-row stores are visible to P5's loads; region A is dead from here

  // ---- P5: preference lists (KAS:62, 202-239) ------------------------------------------------
  // rotation offsets idx_m = Math.abs(hash) % m for every set size m (KAS:190, via KAS:267)
  int32_t idxm[W + 1];
#pragma unroll
  for (int m = 1; m <= W; ++m) idxm[m] = java_abs_mod(hash, m);
  idxm[0] = 0;
  for (int32_t i = lane; i < N; i += 64) L.owner[i] = 0u;
  kasw::sync();
  uint32_t epoch = 0;
  bool hash_fail = false;
  uint64_t digest = 0;
  for (int32_t tile = 0; tile < nt; ++tile) {
    const int32_t p = (tile << 6) + lane;
    const bool active = p < P;
    int32_t h[W];
    int32_t Lp = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      h[k] = 0x7fffffff;
      if (active && k < ow) {
        const int32_t v = out[(int64_t)p * ow + k];
        if (v >= 0) { h[k] = v; Lp += 1; }
      }
    }
    // Sets.newTreeSet(preferenceList) (KAS:228): ascending node index == ascending broker id
#pragma unroll
    for (int pass = 0; pass < W; ++pass) {
#pragma unroll
      for (int k = (pass & 1); k + 1 < W; k += 2) {
        const int32_t lo = h[k] < h[k + 1] ? h[k] : h[k + 1];
        const int32_t hi = h[k] < h[k + 1] ? h[k + 1] : h[k];
        h[k] = lo; h[k + 1] = hi;
      }
    }
    // KAS:190 index error: some set size m <= L has a negative rotation offset
    bool bad = false;
#pragma unroll
    for (int m = 1; m <= W; ++m) bad = bad || (m <= Lp && idxm[m] < 0);
    if (kasw::ballot(bad) != 0) { hash_fail = true; break; }

    bool pending = active && Lp > 0;
    int32_t lst[W];
#pragma unroll
    for (int k = 0; k < W; ++k) lst[k] = -1;
    for (;;) {
      if (kasw::ballot(pending) == 0) break;
      epoch += 1;
      st[6] += 1;
      const uint32_t key = (epoch << 6) | (uint32_t)(63 - lane);
      if (pending) {
#pragma unroll
        for (int k = 0; k < W; ++k) if (k < Lp) kasw::lds_atomic_max(&L.owner[h[k]], key);
      }
      kasw::sync();
      bool ready = pending;
#pragma unroll
      for (int k = 0; k < W; ++k) if (k < Lp) ready = ready && (L.owner[pending ? h[k] : 0] == key);
      if (ready) {
        int32_t c[W][W];
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
          for (int r = 0; r < W; ++r)
            c[k][r] = (k < Lp && r < Lp) ? L.cnt[h[k] * W + r] : 0;
        uint32_t alive = (1u << Lp) - 1u;                 // positions of the sorted set still in nodeSet
        int32_t m = Lp;
#pragma unroll
        for (int r = 0; r < W; ++r) {
          if (r < Lp) {
            // getLeastSeenNodeForReplicaId (KAS:263-278): visit order[j] = S[(j + m - idx) % m]
            const int32_t off = m - sel<W + 1>(idxm, m);
            int32_t best_pos = -1, best_cnt = 0;
#pragma unroll
            for (int jj = 0; jj < W; ++jj) {
              if (jj < m) {
                int32_t q = jj + off; if (q >= m) q -= m;   // rank inside the remaining set
                // position of the q-th alive element
                int32_t pos = -1, seen = 0;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                  const bool al = (alive >> k) & 1u;
                  if (al && seen == q && pos < 0) pos = k;
                  seen += al ? 1 : 0;
                }
                int32_t cv = 0;
#pragma unroll
                for (int k = 0; k < W; ++k) cv = (pos == k) ? c[k][r] : cv;
                if (best_pos < 0 || cv < best_cnt) { best_pos = pos; best_cnt = cv; }   // KAS:270
              }
            }
            alive &= ~(1u << best_pos);                    // nodeSet.remove (KAS:232)
            m -= 1;
            const int32_t node = sel<W>(h, best_pos);
            lst[r] = node;
            L.cnt[node * W + r] = best_cnt + 1;            // updateCountersFromList (KAS:254-261)
          }
        }
        pending = false;
      }
      kasw::sync();
    }
    if (active) {
#pragma unroll
      for (int k = 0; k < W; ++k) {
        if (k < ow) {
          const int32_t node = lst[k];
          const int32_t id = node >= 0 ? g_node_id[node] : -1;
          out[(int64_t)p * ow + k] = id;
          if (node >= 0) digest += kas_digest_cell(topic_k, (uint32_t)p, (uint32_t)k, id);
        }
      }
    }
  }
  { const int64_t now = kasw::clock_ticks(); st[3] += now - tmark; tmark = now; }
  if (hash_fail) { res.status = KAS_FAIL_HASH_INDEX; return res; }
  res.moved_replicas = kasw::wave_sum(moved_r);
  res.moved_partitions = kasw::wave_sum(moved_p);
  res.digest = digest;
  return res;
}

// ---------------------------------------------------------------------------------------------
// One scenario: the per-topic loop of KAG:173-184 against one Context (KTA:19-23).
// ---------------------------------------------------------------------------------------------
template <int W>
KAS_DEV void solve_scenario(const KasLaunch& a, int32_t s, unsigned char* lds_raw) {
  const int lane = kasw::lane();
  const kas_scenario_desc sd = a.scen[s];
  const int32_t N = sd.n_nodes;
  const KasLds lay = kas_lds_layout(a.n_max, W, a.idmap_entries, a.need_bsearch);
  LdsView L;
  L.cnt = (int32_t*)(lds_raw + lay.off_cnt);
  L.owner = (uint32_t*)(lds_raw + lay.off_owner);
  L.load = (int32_t*)(lds_raw + lay.off_load);
  L.rack = (int16_t*)(lds_raw + lay.off_rack);
  L.live = (int16_t*)(lds_raw + lay.off_live);
  L.idmap = (int16_t*)(lds_raw + lay.off_idmap);
  L.ids = (int32_t*)(lds_raw + lay.off_ids);
  L.ring_p = (int32_t*)(lds_raw + lay.off_ring);
  L.ring_meta = L.ring_p + KAS_RING_CAP;
  L.ring_rack = (int16_t*)(L.ring_meta + KAS_RING_CAP);

  const int32_t* g_node_id = a.node_id + sd.node_off;
  const int32_t* g_node_rack = a.node_rack + sd.node_off;
  const bool has_ctx = sd.ctx_off >= 0 && sd.ctx_width > 0;
  int32_t* g_ctx = has_ctx ? a.ctx + sd.ctx_off : nullptr;
  const int32_t ctxw = sd.ctx_width;

  int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int64_t t_begin = kasw::clock_ticks();
  // node table checks (strictly ascending, non-negative ids; racks in int16 range) and the
  // Context counters (KAS:360-369) into LDS
  bool bad = false;
  for (int32_t i = lane; i < N; i += 64) {
    const int32_t id = g_node_id[i];
    const int32_t prev = i > 0 ? g_node_id[i - 1] : -1;
    const int32_t rk = g_node_rack[i];
    bad = bad || id <= prev || rk < 0 || rk > 32767;
#pragma unroll
    for (int r = 0; r < W; ++r)
      L.cnt[i * W + r] = (has_ctx && r < ctxw) ? g_ctx[(int64_t)i * ctxw + r] : 0;
  }
  const bool nodes_bad = kasw::ballot(bad) != 0;
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
  int32_t scen_status = KAS_OK, fail_topic = -1, fail_part = -1;
  int32_t moved_r = 0, moved_p = 0;
  uint64_t digest = 0;
  for (int32_t k = 0; k < sd.topic_count; ++k) {
    const int32_t ti = sd.topic_begin + k;
    const kas_topic_desc td = a.topics[ti];
    TopicOutcome o;
    o.status = KAS_OK; o.fail_partition = -1; o.moved_replicas = 0; o.moved_partitions = 0; o.digest = 0;
    if (scen_status != KAS_OK) o.status = KAS_SKIPPED;                 // KAG:173-184 aborted
    else if (nodes_bad) o.status = KAS_FAIL_BAD_NODES;
    else if (!(td.rf > 0)) o.status = KAS_FAIL_RF_NOT_POSITIVE;        // KTA:65-66
    else if (!(td.rf <= N)) o.status = KAS_FAIL_RF_GT_BROKERS;         // KTA:67-69
    else o = solve_topic<W>(a, td, (uint32_t)k, L, nm, g_node_id, g_node_rack, accmask, st);
    if (o.status != KAS_OK) {
      // nothing is returned for a failed topic: its rows are all padding
      int32_t* out = a.out + td.out_off;
      const int64_t cells = (int64_t)td.n_partitions * td.out_width;
      for (int64_t i = lane; i < cells; i += 64) out[i] = -1;
      o.moved_replicas = 0; o.moved_partitions = 0; o.digest = 0;
      if (scen_status == KAS_OK) { scen_status = o.status; fail_topic = k; fail_part = o.fail_partition; }
    }
    if (lane == 0) {
      kas_topic_result tr;
      tr.status = o.status; tr.fail_partition = o.fail_partition;
      tr.moved_replicas = o.moved_replicas; tr.moved_partitions = o.moved_partitions;
      a.topic_results[ti] = tr;
    }
    moved_r += o.moved_replicas; moved_p += o.moved_partitions;
    digest += o.digest;
    kasw::sync();
  }
  if (has_ctx) {
    for (int32_t i = lane; i < N; i += 64)
#pragma unroll
      for (int r = 0; r < W; ++r)
        if (r < ctxw) g_ctx[(int64_t)i * ctxw + r] = L.cnt[i * W + r];
  }
  const uint64_t dsum = kasw::wave_sum_u64(digest);
  if (lane == 0) {
    kas_scenario_result sr;
    sr.status = scen_status; sr.fail_topic = fail_topic; sr.fail_partition = fail_part;
    sr.moved_replicas = moved_r; sr.moved_partitions = moved_p; sr.reserved = 0;
    sr.digest = dsum;
    a.scenario_results[s] = sr;
    if (a.stats) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a.stats[(int64_t)s * 8 + i] = st[i];
    }
  }
}

}  // namespace kas
