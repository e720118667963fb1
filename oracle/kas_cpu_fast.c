/*
 * kas_cpu_fast.c — CPU BASELINE B2 ("optimised CPU").
 * TEST / BENCH INFRASTRUCTURE, NOT PRODUCT: only tests/ and bench.py's cpu_baseline leg load it.
 *
 * The same path as kas_oracle.c —
 *     KafkaTopicAssigner.generateAssignment          (KTA = KafkaTopicAssigner.java:42-72)
 *     KafkaAssignmentStrategy.getRackAwareAssignment (KAS = KafkaAssignmentStrategy.java:40-63)
 * — with the SAME results (tests/test_cpu_fast.py diffs every output against the oracle), but
 * written the way a CPU programmer would after profiling the reference:
 *   * flat int32 arrays, one allocation block per topic, no per-row containers;
 *   * broker id -> node index through a direct table when the id range is dense (KAS:119 is a
 *     TreeMap.get per replica in the reference);
 *   * assignOrphans (KAS:162-186) walks only the nodes that are not full: a full node never
 *     accepts again (Node.canAccept, KAS:320-324), and the reference spends > 99 % of its probes
 *     on them.  The list of non-full nodes, in processing order (KAS:188-200),
 *     is compacted as nodes fill up;
 *   * computePreferenceLists (KAS:202-239) keeps count[node][replica] in one dense table and
 *     evaluates "first strictly smaller count in rotated order" (KAS:263-278) as the minimum of
 *     (count, visit position) without building the rotated array.
 * It is what the GPU path should be compared with when the question is "what can a host do",
 * whereas kas_oracle.c keeps the reference's cost model.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "kas_abi.h"
#include "kas_batch_loop.h"

#if defined(__GNUC__)
#define KAS_FAST_API __attribute__((visibility("default")))
#else
#define KAS_FAST_API
#endif

#define FAST_DIRECT_TABLE_LIMIT (1 << 22)   /* ids spanning more than this use binary search */

/* getMaxReplicasPerNode (KAS:65-71): int product (wraps), double divide, ceil, (int) saturates */
static int32_t fast_cap(int32_t n_nodes, int32_t n_partitions, int32_t rf) {
  const int32_t prod = (int32_t)((uint32_t)n_partitions * (uint32_t)rf);
  const double c = ceil((double)prod / (double)n_nodes);
  if (c >= 2147483647.0) return INT32_MAX;
  if (c <= -2147483648.0) return INT32_MIN;
  return (int32_t)c;
}

/* Math.abs(hash) % m with Java semantics (KAS:190): negative only for Integer.MIN_VALUE */
static int32_t fast_abs_mod(int32_t hash, int32_t m) {
  const int32_t a = (hash == INT32_MIN) ? INT32_MIN : (hash < 0 ? -hash : hash);
  return a % m;
}

typedef struct {
  int32_t N, min_id;
  const int32_t* node_id;
  int32_t* direct;       /* [range] id - min_id -> node index or -1; NULL = binary search */
  int64_t range;
} fast_idmap;

static inline int32_t fast_lookup(const fast_idmap* m, int32_t id) {
  if (m->direct) {
    const int64_t d = (int64_t)id - m->min_id;
    return (d >= 0 && d < m->range) ? m->direct[d] : -1;
  }
  int32_t lo = 0, hi = m->N - 1;
  while (lo <= hi) {
    const int32_t mid = lo + (hi - lo) / 2, v = m->node_id[mid];
    if (v == id) return mid;
    if (v < id) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

KAS_FAST_API
int kas_cpu_fast_solve_topic(int32_t name_hash, int32_t P, const int32_t* part_id,
                             const int32_t* cur, int32_t cur_width, const int32_t* cur_len,
                             const int32_t* in_partitions,
                             int32_t N, const int32_t* node_id, const int32_t* node_rack,
                             int32_t rf, int32_t* counter, int32_t cw,
                             int32_t* out, int32_t out_width,
                             kas_topic_result* res, int64_t* probes) {
  res->status = KAS_OK; res->fail_partition = -1;
  res->moved_replicas = 0; res->moved_partitions = 0;
  const int64_t cells = (int64_t)P * out_width;
  for (int64_t i = 0; i < cells; ++i) out[i] = -1;

  for (int32_t i = 0; i < N; ++i)
    if (node_id[i] < 0 || (i > 0 && node_id[i] <= node_id[i - 1]) || node_rack[i] < 0 || node_rack[i] > 32767) {
      res->status = KAS_FAIL_BAD_NODES; return res->status;
    }
  if (!(rf > 0)) { res->status = KAS_FAIL_RF_NOT_POSITIVE; return res->status; }   /* KTA:65-66 */
  if (!(rf <= N)) { res->status = KAS_FAIL_RF_GT_BROKERS; return res->status; }    /* KTA:67-69 */

  int32_t n_in = P;
  if (in_partitions) { n_in = 0; for (int32_t p = 0; p < P; ++p) n_in += in_partitions[p] ? 1 : 0; }
  const int32_t cap = fast_cap(N, n_in, rf);                                       /* KAS:45 */
  int32_t hw = cur_width > rf ? cur_width : rf;
  if (hw < 1) hw = 1;

  /* one block: load[N] live[N] hold[P][hw] hrk[P][hw] hcnt[P] */
  const size_t Pn = (size_t)(P > 0 ? P : 1), Nn = (size_t)(N > 0 ? N : 1);
  kas_arena* own_arena = kas_tls_arena == NULL ? kas_arena_acquire() : NULL;   /* (a direct caller outside the batch loop) */
  const size_t scratch_mark = kas_scratch_mark();          /* the calling thread's arena (kas_batch_loop.h) */
  int32_t* block = (int32_t*)kas_scratch_alloc(sizeof(int32_t) * (2 * Nn + 2 * Pn * (size_t)hw + Pn), 0);
  int32_t* load = block;
  int32_t* live = load + Nn;
  int32_t* hold = live + Nn;
  int32_t* hrk = hold + Pn * (size_t)hw;
  int32_t* hcnt = hrk + Pn * (size_t)hw;
  memset(load, 0, sizeof(int32_t) * Nn);
  memset(hcnt, 0, sizeof(int32_t) * Pn);

  fast_idmap im;
  im.N = N; im.node_id = node_id; im.direct = NULL; im.min_id = N > 0 ? node_id[0] : 0;
  im.range = N > 0 ? (int64_t)node_id[N - 1] - node_id[0] + 1 : 0;
  if (N > 0 && im.range <= FAST_DIRECT_TABLE_LIMIT && im.range <= 64 * (int64_t)N + 1024) {
    im.direct = (int32_t*)kas_scratch_alloc(sizeof(int32_t) * (size_t)im.range, 0);
    for (int64_t i = 0; i < im.range; ++i) im.direct[i] = -1;
    for (int32_t i = 0; i < N; ++i) im.direct[node_id[i] - im.min_id] = i;
  }

  /* ---- P2 fillNodesFromAssignment (KAS:101-131): (replica index, partition) ascending ------- */
  for (int32_t r = 0; r < cur_width; ++r) {
    for (int32_t p = 0; p < P; ++p) {
      if (cur_len && r >= cur_len[p]) continue;
      const int32_t n = fast_lookup(&im, cur[(int64_t)p * cur_width + r]);
      if (n < 0 || load[n] >= cap) continue;                        /* KAS:119-120, 322      */
      const int32_t c = hcnt[p], rk = node_rack[n];
      int32_t* h = hold + (size_t)p * hw;
      int32_t* hr = hrk + (size_t)p * hw;
      int ok = 1;
      for (int32_t k = 0; k < c; ++k) ok &= (h[k] != n) & (hr[k] != rk);   /* KAS:321, 346-348 */
      if (!ok) continue;
      h[c] = n; hr[c] = rk; hcnt[p] = c + 1; load[n] += 1;          /* KAS:326-331, 350-354  */
    }
  }

  /* ---- P3 + P4: orphans (KAS:133-160) first-fit over the non-full nodes (KAS:162-186) ------- */
  int32_t fail_row = -1;
  const int32_t idxN = fast_abs_mod(name_hash, N);                  /* KAS:168 -> KAS:190     */
  if (idxN < 0) {
    res->status = KAS_FAIL_HASH_INDEX;
  } else {
    /* order[j] = sorted[(j - idx) mod N] (KAS:191-198) */
    int32_t n_live = 0, dead = 0;
    const int32_t start = (N - idxN) % N;
    for (int32_t j = 0; j < N; ++j) {
      int32_t n = j + start; if (n >= N) n -= N;
      if (load[n] < cap) live[n_live++] = n;
    }
    for (int32_t p = 0; p < P && fail_row < 0; ++p) {               /* TreeMap order KAS:172  */
      if (in_partitions && !in_partitions[p]) continue;             /* KAS:150                */
      int32_t remaining = rf - hcnt[p];
      if (remaining <= 0) continue;
      int32_t* h = hold + (size_t)p * hw;
      int32_t* hr = hrk + (size_t)p * hw;
      for (int32_t j = 0; j < n_live && remaining > 0; ++j) {       /* from order[0] KAS:175  */
        const int32_t n = live[j];
        if (probes) ++*probes;
        if (load[n] >= cap) continue;                               /* became full meanwhile  */
        const int32_t c = hcnt[p], rk = node_rack[n];
        int ok = 1;
        for (int32_t k = 0; k < c; ++k) ok &= (h[k] != n) & (hr[k] != rk);
        if (!ok) continue;
        h[c] = n; hr[c] = rk; hcnt[p] = c + 1;
        if (++load[n] >= cap) ++dead;
        --remaining;
      }
      if (remaining != 0) { fail_row = p; break; }                  /* KAS:183-184            */
      if (dead > 8 && dead * 4 > n_live) {                          /* drop the full nodes    */
        int32_t w = 0;
        for (int32_t j = 0; j < n_live; ++j) if (load[live[j]] < cap) live[w++] = live[j];
        n_live = w; dead = 0;
      }
    }
    if (fail_row >= 0) {
      res->status = KAS_FAIL_UNASSIGNABLE;
      res->fail_partition = part_id ? part_id[fail_row] : fail_row;
    }
  }

  /* ---- P5 computePreferenceLists (KAS:202-239) + movement ------------------------------------ */
  if (res->status == KAS_OK) {
    int32_t idxm[KAS_MAX_WIDTH * 2 + 1];
    idxm[0] = 0;
    for (int32_t m = 1; m <= hw; ++m) idxm[m] = fast_abs_mod(name_hash, m);
    int32_t moved_r = 0, moved_p = 0;
    for (int32_t p = 0; p < P && res->status == KAS_OK; ++p) {
      const int32_t L = hcnt[p];
      int32_t* row = out + (int64_t)p * out_width;
      int32_t set[KAS_MAX_WIDTH * 2], list[KAS_MAX_WIDTH * 2];
      const int32_t* h = hold + (size_t)p * hw;
      for (int32_t k = 0; k < L; ++k) {                             /* Sets.newTreeSet KAS:228 */
        int32_t v = h[k], j = k - 1;
        while (j >= 0 && set[j] > v) { set[j + 1] = set[j]; --j; }
        set[j + 1] = v;
      }
      int32_t sz = L;
      for (int32_t r = 0; r < L; ++r) {                             /* KAS:229-233             */
        const int32_t idx = idxm[sz];
        if (idx < 0) { res->status = KAS_FAIL_HASH_INDEX; break; }  /* KAS:190 index error     */
        /* sorted element i is visited at position (i + idx) % sz; strictly smaller count wins,
         * earlier visit wins ties (KAS:263-278) */
        int32_t best_i = 0;
        int64_t best_key = INT64_MAX;
        for (int32_t i = 0; i < sz; ++i) {
          int32_t pos = i + idx; if (pos >= sz) pos -= sz;
          const int64_t key = (int64_t)counter[(int64_t)set[i] * cw + r] * 256 + pos;
          if (key < best_key) { best_key = key; best_i = i; }
        }
        list[r] = set[best_i];
        for (int32_t i = best_i; i + 1 < sz; ++i) set[i] = set[i + 1];   /* nodeSet.remove KAS:232 */
        --sz;
      }
      if (res->status != KAS_OK) break;
      for (int32_t r = 0; r < L; ++r) {                             /* KAS:254-261             */
        counter[(int64_t)list[r] * cw + r] += 1;
        row[r] = node_id[list[r]];
      }
      /* movement (include/kas_abi.h): gained = new brokers not in cur; differ = a cur broker gone */
      const int32_t* c = cur + (int64_t)p * cur_width;
      const int32_t clen = cur_len ? cur_len[p] : cur_width;
      int32_t gained = 0, differ = 0;
      for (int32_t k = 0; k < L; ++k) {
        int found = 0;
        for (int32_t q = 0; q < clen; ++q) found |= c[q] == row[k];
        gained += !found;
      }
      for (int32_t q = 0; q < clen; ++q) {
        int found = 0;
        for (int32_t k = 0; k < L; ++k) found |= row[k] == c[q];
        differ |= !found;
      }
      moved_r += gained;
      moved_p += (gained || differ) ? 1 : 0;
    }
    res->moved_replicas = moved_r; res->moved_partitions = moved_p;
  }
  if (res->status != KAS_OK) {
    for (int64_t i = 0; i < cells; ++i) out[i] = -1;
    res->moved_replicas = 0; res->moved_partitions = 0;
  }
  kas_scratch_release(scratch_mark);
  if (own_arena) kas_arena_return(own_arena);
  return res->status;
}

KAS_FAST_API
int kas_cpu_fast_solve_batch(const kas_batch_desc* b, const kas_tables* t) {
  const int rc = kas_loop_batch(b, t, kas_cpu_fast_solve_topic, 1);
  return rc < 0 ? rc : KAS_E_OK;
}

/* scenario-parallel on n_threads host threads inside this one call (<= 0: every hardware thread);
 * returns the number of threads used, or a negative KAS_E_* code */
KAS_FAST_API
int kas_cpu_fast_solve_batch_mt(const kas_batch_desc* b, const kas_tables* t, int n_threads) {
  return kas_loop_batch(b, t, kas_cpu_fast_solve_topic, n_threads);
}

KAS_FAST_API
int kas_cpu_fast_host_threads(void) { return kas_loop_host_threads(); }

/* seconds n_threads threads take for a fixed register-only loop each (kas_batch_loop.h) */
KAS_FAST_API
double kas_cpu_fast_parallelism_probe(int n_threads, uint64_t iters) { return kas_loop_parallelism_probe(n_threads, iters); }

KAS_FAST_API
int kas_cpu_fast_abi_version(void) { return KAS_ABI_VERSION; }
