/*
 * kas_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * A plain-C restatement of the reference algorithm for the one hot path this repository
 * accelerates:
 *     KafkaTopicAssigner.generateAssignment          (KTA  = KafkaTopicAssigner.java:42-72)
 *     KafkaAssignmentStrategy.getRackAwareAssignment (KAS  = KafkaAssignmentStrategy.java:40-63)
 * Every function below cites the reference lines it follows.  It deliberately keeps the
 * reference's control flow and cost model (e.g. assignOrphans rescans the node order from
 * its start for every orphan, KAS:175-176); only the containers differ: a TreeSet<Integer>
 * of partitions per Node/Rack becomes a per-partition holder list, which answers the same
 * three questions canAccept asks (KAS:320-324, 346-348).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product library (kafka-assigner_amd/csrc) never links or calls it.
 *
 * Parity pin: the reference cannot be compiled here (no JDK, un-vendored Maven deps), and its
 * own tests (KafkaTopicAssignerTest.java) hold invariant assertions plus ONE exact pin
 * (testReplacement: new[0] == [10,11]).  This oracle is checked against all of those
 * assertions, against an independent line-by-line Python restatement (oracle/literal_ref.py)
 * and against the hand-traced vectors of SURVEY.md Appendix B (tests/golden/).  Exact list
 * ORDER beyond that one pin is pinned by source reading only.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>

#include "kas_abi.h"
#include "kas_batch_loop.h"

#if defined(__GNUC__)
#define KAS_ORACLE_API __attribute__((visibility("default")))
#else
#define KAS_ORACLE_API
#endif

/* ------------------------------------------------------------------------------------------
 * The node/rack model (KAS:307-355).  load[n] = |Node.assignedPartitions|; the partitions a
 * node or rack holds are recorded from the partition's side: holders[p][0..hcnt[p]) are the
 * node indices currently holding row p.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t N;
  const int32_t* node_id;    /* strictly ascending == TreeMap<Integer,Node> order (KAS:78)   */
  const int32_t* node_rack;  /* equal index <=> equal rack id string (KAS:81-94)             */
  int32_t cap;               /* Node.capacity (KAS:309, from KAS:45)                         */
  int32_t* load;             /* [N]                                                          */
  int32_t P;
  int32_t hw;                /* holder list capacity per row                                 */
  int32_t* holders;          /* [P][hw] node index                                           */
  int32_t* hcnt;             /* [P]                                                          */
} model_t;

/* Node.canAccept (KAS:320-324) + Rack.canAccept (KAS:346-348). */
static int can_accept(const model_t* m, int32_t n, int32_t p) {
  const int32_t* h = m->holders + (int64_t)p * m->hw;
  int32_t c = m->hcnt[p];
  for (int32_t k = 0; k < c; ++k)                 /* !assignedPartitions.contains(partition) */
    if (h[k] == n) return 0;
  if (!(m->load[n] < m->cap)) return 0;           /* assignedPartitions.size() < capacity     */
  for (int32_t k = 0; k < c; ++k)                 /* rack.canAccept(partition)                */
    if (m->node_rack[h[k]] == m->node_rack[n]) return 0;
  return 1;
}

/* Node.accept (KAS:326-331) + Rack.accept (KAS:350-354). */
static void accept(model_t* m, int32_t n, int32_t p) {
  m->holders[(int64_t)p * m->hw + m->hcnt[p]] = n;
  m->hcnt[p] += 1;
  m->load[n] += 1;
}

/* nodeMap.get(nodeId) (KAS:119): index of id in the ascending node table, or -1. */
static int32_t node_index(const model_t* m, int32_t id) {
  int32_t lo = 0, hi = m->N - 1;
  while (lo <= hi) {
    int32_t mid = lo + (hi - lo) / 2;
    int32_t v = m->node_id[mid];
    if (v == id) return mid;
    if (v < id) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

/* getMaxReplicasPerNode (KAS:65-71): int product first, then double divide and ceil. */
static int32_t get_max_replicas_per_node(int32_t n_nodes, int32_t n_partitions, int32_t rf) {
  int32_t prod = (int32_t)((uint32_t)n_partitions * (uint32_t)rf);   /* KAS:69 int multiply */
  double total = (double)prod;
  double c = ceil(total / (double)n_nodes);                          /* KAS:70              */
  if (c >= 2147483647.0) return INT32_MAX;                           /* Java (int) saturates */
  if (c <= -2147483648.0) return INT32_MIN;
  return (int32_t)c;
}

/* getNodeProcessingOrder (KAS:188-200).  ids_sorted[0..n) ascending; writes order[0..n).
 * Returns 0, or -1 for the negative-index case (hashCode()==Integer.MIN_VALUE, KAS:190). */
static int get_node_processing_order(int32_t name_hash, const int32_t* ids_sorted, int32_t n,
                                     int32_t* order) {
  int32_t a = (name_hash == INT32_MIN) ? INT32_MIN : (name_hash < 0 ? -name_hash : name_hash);
  int32_t index = a % n;                       /* Java %: truncating, sign of the dividend  */
  for (int32_t i = 0; i < n; ++i) {            /* KAS:191-198                               */
    if (index < 0) return -1;                  /* ArrayIndexOutOfBoundsException             */
    order[index] = ids_sorted[i];
    if (++index == n) index = 0;
  }
  return 0;
}

/* fillNodesFromAssignment (KAS:101-131): sweep r = 0,1,2,...; inside a sweep rows ascending;
 * a row takes part while it still has an r-th current replica (KAS:117-127). */
static void fill_nodes_from_assignment(model_t* m, const int32_t* cur, int32_t cur_width,
                                       const int32_t* cur_len) {
  for (int32_t r = 0; r < cur_width; ++r) {
    for (int32_t p = 0; p < m->P; ++p) {
      int32_t len = cur_len ? cur_len[p] : cur_width;
      if (r >= len) continue;                                  /* iterator exhausted         */
      int32_t n = node_index(m, cur[(int64_t)p * cur_width + r]);
      if (n >= 0 && can_accept(m, n, p)) accept(m, n, p);      /* KAS:119-124                */
    }
  }
}

/* getOrphanedReplicas (KAS:133-160): orphans[p] = rf - assigned(p) when positive, for p in
 * `partitions` ascending.  Returns the number of orphaned rows; rem[p] holds the counts. */
static int32_t get_orphaned_replicas(const model_t* m, const int32_t* in_partitions, int32_t rf,
                                     int32_t* rem) {
  int32_t n_orphans = 0;
  for (int32_t p = 0; p < m->P; ++p) {
    rem[p] = 0;
    if (in_partitions && !in_partitions[p]) continue;          /* KAS:150 iterates partitions */
    int32_t remaining = rf - m->hcnt[p];                       /* KAS:151-154                */
    if (remaining > 0) { rem[p] = remaining; ++n_orphans; }    /* KAS:155-157                */
  }
  return n_orphans;
}

/* assignOrphans (KAS:162-186).  Returns -1 ok, -2 for the KAS:168/190 index error, else the
 * row index of the first partition that could not be fully assigned (KAS:183-184). */
static int32_t assign_orphans(model_t* m, int32_t name_hash, const int32_t* rem,
                              int32_t* idx_sorted, int32_t* order, int64_t* probes) {
  for (int32_t i = 0; i < m->N; ++i) idx_sorted[i] = i;        /* nodeMap.keySet() ascending */
  if (get_node_processing_order(name_hash, idx_sorted, m->N, order) != 0) return -2;
  for (int32_t p = 0; p < m->P; ++p) {                         /* TreeMap order (KAS:172)    */
    int32_t remaining = rem[p];
    if (remaining <= 0) continue;
    for (int32_t j = 0; j < m->N && remaining > 0; ++j) {      /* from order[0] (KAS:175)    */
      int32_t n = order[j];
      if (probes) ++*probes;
      if (can_accept(m, n, p)) { accept(m, n, p); --remaining; }
    }
    if (remaining != 0) return p;                              /* KAS:183-184                */
  }
  return -1;
}

static void sort_small(int32_t* a, int32_t n) {
  for (int32_t i = 1; i < n; ++i) {
    int32_t v = a[i], j = i - 1;
    while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
    a[j + 1] = v;
  }
}

/* computePreferenceLists (KAS:202-239) with PreferenceListOrderTracker (KAS:244-302).
 * counter[n*cw + r] is Context.counter.get(node).get(r) (missing == 0, KAS:289-300).
 * Writes out rows as broker ids, -1 padded.  Returns 0, or -2 on the KAS:190 index error. */
static int compute_preference_lists(const model_t* m, int32_t name_hash, int32_t* counter,
                                    int32_t cw, int32_t* out, int32_t out_width) {
  int32_t set[KAS_MAX_WIDTH * 2], order[KAS_MAX_WIDTH * 2], list[KAS_MAX_WIDTH * 2];
  for (int32_t p = 0; p < m->P; ++p) {
    int32_t* row = out + (int64_t)p * out_width;
    for (int32_t k = 0; k < out_width; ++k) row[k] = -1;
    int32_t L = m->hcnt[p];
    if (L == 0) continue;                         /* not a key of unorderedPreferences       */
    for (int32_t k = 0; k < L; ++k) set[k] = m->holders[(int64_t)p * m->hw + k];
    sort_small(set, L);                           /* Sets.newTreeSet(preferenceList) KAS:228 */
    int32_t sz = L;
    for (int32_t replica = 0; replica < L; ++replica) {       /* KAS:229                     */
      /* getLeastSeenNodeForReplicaId (KAS:263-278) */
      if (get_node_processing_order(name_hash, set, sz, order) != 0) return -2;
      int32_t min_node = -1, min_count = 0;
      for (int32_t j = 0; j < sz; ++j) {
        int32_t count = counter[(int64_t)order[j] * cw + replica];
        if (min_node < 0 || count < min_count) { min_count = count; min_node = order[j]; }
      }
      int32_t w = 0;                              /* nodeSet.remove(nodeToSelect) KAS:232    */
      for (int32_t j = 0; j < sz; ++j) if (set[j] != min_node) set[w++] = set[j];
      sz = w;
      list[replica] = min_node;
    }
    for (int32_t r = 0; r < L; ++r) {             /* updateCountersFromList KAS:254-261      */
      counter[(int64_t)list[r] * cw + r] += 1;
      row[r] = m->node_id[list[r]];
    }
  }
  return 0;
}

static int contains(const int32_t* a, int32_t n, int32_t v) {
  for (int32_t i = 0; i < n; ++i) if (a[i] == v) return 1;
  return 0;
}

/* moved_replicas / moved_partitions as defined in include/kas_abi.h. */
static void movement(const int32_t* cur, int32_t cur_width, const int32_t* cur_len,
                     const int32_t* out, int32_t out_width, int32_t P,
                     int32_t* moved_replicas, int32_t* moved_partitions) {
  int32_t mr = 0, mp = 0;
  for (int32_t p = 0; p < P; ++p) {
    const int32_t* c = cur + (int64_t)p * cur_width;
    const int32_t* o = out + (int64_t)p * out_width;
    int32_t clen = cur_len ? cur_len[p] : cur_width;
    int32_t olen = 0;
    while (olen < out_width && o[olen] != -1) ++olen;
    int32_t gained = 0, differ = 0;
    for (int32_t k = 0; k < olen; ++k) if (!contains(c, clen, o[k])) ++gained;
    for (int32_t k = 0; k < clen; ++k) if (!contains(o, olen, c[k])) differ = 1;
    mr += gained;
    if (gained || differ) ++mp;
  }
  *moved_replicas = mr;
  *moved_partitions = mp;
}

static void fill_minus_one(int32_t* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = -1;
}

/*
 * One generateAssignment call (KTA:42-72 -> KAS:40-63) on flat tables.
 *   counter: Context counters [N][cw], updated in place (only when the topic succeeds, as in
 *            the reference where P5 is the only phase that touches the Context).
 *   probes:  optional count of canAccept calls made by assignOrphans (cost-model evidence).
 * Returns the topic status; on failure `out` is filled with -1.
 */
KAS_ORACLE_API
int kas_oracle_solve_topic(int32_t name_hash, int32_t P, const int32_t* part_id,
                           const int32_t* cur, int32_t cur_width, const int32_t* cur_len,
                           const int32_t* in_partitions,
                           int32_t N, const int32_t* node_id, const int32_t* node_rack,
                           int32_t rf, int32_t* counter, int32_t cw,
                           int32_t* out, int32_t out_width,
                           kas_topic_result* res, int64_t* probes) {
  res->status = KAS_OK; res->fail_partition = -1;
  res->moved_replicas = 0; res->moved_partitions = 0;
  fill_minus_one(out, (int64_t)P * out_width);

  for (int32_t i = 0; i < N; ++i) {
    if (node_id[i] < 0 || (i > 0 && node_id[i] <= node_id[i - 1]) ||
        node_rack[i] < 0 || node_rack[i] > 32767) {
      res->status = KAS_FAIL_BAD_NODES; return res->status;
    }
  }
  if (!(rf > 0)) { res->status = KAS_FAIL_RF_NOT_POSITIVE; return res->status; }   /* KTA:65 */
  if (!(rf <= N)) { res->status = KAS_FAIL_RF_GT_BROKERS; return res->status; }    /* KTA:67 */

  int32_t n_in = 0;
  for (int32_t p = 0; p < P; ++p) n_in += (!in_partitions || in_partitions[p]) ? 1 : 0;

  model_t m;
  m.N = N; m.node_id = node_id; m.node_rack = node_rack; m.P = P;
  m.cap = get_max_replicas_per_node(N, n_in, rf);                                  /* KAS:45 */
  m.hw = (cur_width > rf ? cur_width : rf); if (m.hw < 1) m.hw = 1;
  kas_arena* own_arena = kas_tls_arena == NULL ? kas_arena_acquire() : NULL;   /* (a direct caller outside the batch loop) */
  /* scratch from the calling thread's arena (kas_batch_loop.h): no malloc / free per topic */
  const size_t scratch_mark = kas_scratch_mark();
  m.load = (int32_t*)kas_scratch_alloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1), 1);          /* KAS:46 */
  m.holders = (int32_t*)kas_scratch_alloc(sizeof(int32_t) * (size_t)(P > 0 ? P : 1) * m.hw, 0);
  m.hcnt = (int32_t*)kas_scratch_alloc(sizeof(int32_t) * (size_t)(P > 0 ? P : 1), 1);
  int32_t* rem = (int32_t*)kas_scratch_alloc(sizeof(int32_t) * (size_t)(P > 0 ? P : 1), 0);
  int32_t* idx_sorted = (int32_t*)kas_scratch_alloc(sizeof(int32_t) * (size_t)N, 0);
  int32_t* order = (int32_t*)kas_scratch_alloc(sizeof(int32_t) * (size_t)N, 0);

  fill_nodes_from_assignment(&m, cur, cur_width, cur_len);                         /* KAS:49 */
  get_orphaned_replicas(&m, in_partitions, rf, rem);                               /* KAS:52 */
  int32_t bad = assign_orphans(&m, name_hash, rem, idx_sorted, order, probes);     /* KAS:56 */
  if (bad == -2) {
    res->status = KAS_FAIL_HASH_INDEX;
  } else if (bad >= 0) {
    res->status = KAS_FAIL_UNASSIGNABLE;
    res->fail_partition = part_id ? part_id[bad] : bad;
  } else {
    /* A KAS:190 failure inside P5 leaves the counters partially updated, exactly as the
     * exception would.  The reference's Context is unusable after that (the CLI run has
     * aborted), so only the status is compared for that case. */
    if (compute_preference_lists(&m, name_hash, counter, cw, out, out_width) != 0) /* KAS:62 */
      res->status = KAS_FAIL_HASH_INDEX;
  }
  if (res->status != KAS_OK) {
    fill_minus_one(out, (int64_t)P * out_width);
  } else {
    movement(cur, cur_width, cur_len, out, out_width, P,
             &res->moved_replicas, &res->moved_partitions);
  }
  kas_scratch_release(scratch_mark);
  if (own_arena) kas_arena_return(own_arena);
  return res->status;
}

/*
 * Whole batch with the semantics of kas_solve_host (include/kas_abi.h); the per-scenario topic
 * loop (one Context per scenario, first failure skips the rest: KAG:173-184) is in
 * kas_batch_loop.h.  `tables` holds HOST pointers.  Returns 0, or a negative KAS_E_* code for
 * malformed descriptors.
 */
KAS_ORACLE_API
int kas_oracle_solve_batch(const kas_batch_desc* b, const kas_tables* t) {
  const int rc = kas_loop_batch(b, t, kas_oracle_solve_topic, 1);
  return rc < 0 ? rc : KAS_E_OK;
}

/* The same batch, scenario-parallel on n_threads host threads inside this one call (<= 0: every
 * hardware thread).  Returns the number of threads used, or a negative KAS_E_* code. */
KAS_ORACLE_API
int kas_oracle_solve_batch_mt(const kas_batch_desc* b, const kas_tables* t, int n_threads) {
  return kas_loop_batch(b, t, kas_oracle_solve_topic, n_threads);
}

KAS_ORACLE_API
int kas_oracle_host_threads(void) { return kas_loop_host_threads(); }

KAS_ORACLE_API
int kas_oracle_abi_version(void) { return KAS_ABI_VERSION; }
