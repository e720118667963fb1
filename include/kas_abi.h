/*
 * kas_abi.h — C ABI of the MI355X batch solver for kafka-assigner's minimal-movement,
 * rack-aware rebalance path.
 *
 * This is the drop-in boundary for exactly one reference call site:
 *
 *   KafkaTopicAssigner.generateAssignment            KafkaTopicAssigner.java:42-72
 *     -> KafkaAssignmentStrategy.getRackAwareAssignment
 *                                                     KafkaAssignmentStrategy.java:40-63
 *
 * Everything the reference passes as boxed Java collections is passed here as flat,
 * caller-owned int32 tables:
 *
 *   reference argument (KafkaAssignmentStrategy.java:40-43)     this ABI
 *   ----------------------------------------------------------  ---------------------------------
 *   String topicName                                            kas_topic_desc.name_hash
 *                                                               (= topicName.hashCode(); only use
 *                                                               of the name is KAS:190)
 *   Map<Integer,List<Integer>> currentAssignment                cur pool: cur[P][cur_width] broker
 *                                                               ids, rows in ascending partition
 *                                                               id order (+ optional cur_len[P])
 *   Map<Integer,String> nodeRackAssignment                      node_rack[N] dense rack index
 *                                                               (rack-less broker = own unique
 *                                                               index, KAS:82-86)
 *   Set<Integer> nodes                                          node_id[N] strictly ascending
 *   Set<Integer> partitions                                     implicit = rows of cur; optional
 *                                                               in_partitions[P] flags for direct
 *                                                               callers that pass a different set
 *   int replicationFactor                                       kas_topic_desc.rf (resolved per
 *                                                               KafkaTopicAssigner.java:49-62)
 *   Context context                                             ctx pool: counter[N][ctx_width]
 *                                                               (KAS:360-369, KAS:244-302)
 *   return Map<Integer,List<Integer>>                           out pool: out[P][out_width] broker
 *                                                               ids in preference order, -1 padded
 *   IllegalStateException (KAS:183-184, KTA:65-69)              kas_topic_result.status/.fail_partition
 *
 * The boundary is generateAssignment-level: the replication-factor preconditions of
 * KafkaTopicAssigner.java:65-69 are applied by the solver (KAS_FAIL_RF_NOT_POSITIVE /
 * KAS_FAIL_RF_GT_BROKERS), i.e. rf outside [1, N] never reaches the KAS:40-63 phases, exactly
 * as no call through KTA:70-71 can carry such an rf.  (getRackAwareAssignment itself has no such
 * checks; callers that bypass KafkaTopicAssigner see the KTA status for those two ranges.)
 *
 * A *scenario* is one cluster snapshot: one broker set + rack map and an ordered list of
 * topics that share one Context (what one `--mode PRINT_REASSIGNMENT` run of
 * KafkaAssignmentGenerator.java:131-187 solves).  A *batch* is many independent scenarios;
 * the GPU solves them concurrently, one wavefront per scenario.
 *
 * Rules: no C++ types, exceptions or longjmp cross this boundary; the library never
 * retains caller pointers after a call returns (a plan copies what it needs); every
 * function returns 0 or a negative KAS_E_* code and kas_last_error() gives thread-local
 * detail.  There is no CPU fallback: without a HIP device the solve entry points fail.
 */
#ifndef KAS_ABI_H
#define KAS_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KAS_ABI_VERSION 5

/* Longest replica list the kernels keep in registers: max(cur_width, rf) <= KAS_MAX_WIDTH. */
#define KAS_MAX_WIDTH 8

/* ---- return codes of the entry points (negative = call failed) ------------------------ */
#define KAS_E_OK            0
#define KAS_E_INVALID_ARG  (-1)  /* NULL pointer, negative size, inconsistent descriptor   */
#define KAS_E_HIP          (-2)  /* a HIP runtime call failed or no gfx950 device present   */
#define KAS_E_UNSUPPORTED  (-3)  /* shape beyond what the kernels hold in LDS / registers   */
#define KAS_E_NOMEM        (-4)

/* ---- per-topic status codes (the reference's exceptions, as data) --------------------- */
#define KAS_OK                   0
#define KAS_FAIL_UNASSIGNABLE    1  /* KAS:183-184 "Partition p could not be fully assigned!";
                                       fail_partition = p                                   */
#define KAS_FAIL_RF_NOT_POSITIVE 2  /* KTA:65-66                                            */
#define KAS_FAIL_RF_GT_BROKERS   3  /* KTA:67-69                                            */
#define KAS_FAIL_HASH_INDEX      4  /* topic.hashCode()==Integer.MIN_VALUE: Math.abs stays
                                       negative -> ArrayIndexOutOfBounds at KAS:190-192      */
#define KAS_FAIL_RF_MISMATCH     5  /* KTA:58-60; raised while resolving rf on the host
                                       (kas_resolve_replication_factor, the mirrors), never by
                                       the kernels                                           */
#define KAS_SKIPPED              6  /* an earlier topic of the same scenario failed; the CLI
                                       run would have aborted (KAG:173-184)                  */
#define KAS_FAIL_BAD_NODES       7  /* node_id[] not strictly ascending (KAS:80 analogue),
                                       a negative node id (-1 is the pad value of out rows;
                                       Kafka broker ids are non-negative), or node_rack[]
                                       outside [0, 32767]                                    */

#define KAS_FAIL_WATCHDOG        8  /* internal error, reported instead of a hung GPU: a wavefront
                                       polled KAS_SPIN_BOUND times (2^25: seconds) for another
                                       wavefront of its workgroup without progress; the scenario's
                                       rows are unspecified.  No input is known to cause it.  (Fill
                                       kernel and order kernels for lists <= 3 wide; the wide form
                                       carries the bound in test builds only)                      */

/* One topic of one scenario.  Offsets are in int32 elements into the named pool. */
typedef struct kas_topic_desc {
  int32_t name_hash;        /* Java String.hashCode() of the topic name (KAS:190)           */
  int32_t n_partitions;     /* P = number of rows of cur (= |keys(currentAssignment)|)      */
  int32_t cur_width;        /* columns of the cur table (longest current replica list)      */
  int32_t rf;               /* replication factor handed to KAS:40-43                       */
  int32_t out_width;        /* columns of the out table, >= max(cur_width, rf)              */
  int32_t reserved;
  int64_t cur_off;          /* cur pool: cur[P][cur_width], row-major, broker ids            */
  int64_t out_off;          /* out pool: out[P][out_width]                                   */
  int64_t cur_len_off;      /* aux pool: int32 len[P] (ragged lists), or -1 = all full width */
  int64_t in_partitions_off;/* aux pool: int32 flag[P] (row is in `partitions`), or -1 = all */
  int64_t part_id_off;      /* aux pool: ascending int32 partition ids[P], or -1 = 0..P-1    */
} kas_topic_desc;

/* One scenario = one broker set and an ordered run of topics sharing one Context. */
typedef struct kas_scenario_desc {
  int32_t n_nodes;          /* N                                                             */
  int32_t topic_begin;      /* first topic (index into the topic descriptor array)           */
  int32_t topic_count;      /* topics solved in this order against one Context               */
  int32_t ctx_width;        /* columns of the Context counter table (0 = no Context in/out)  */
  int64_t node_off;         /* node pools: node_id[N] (strictly ascending) and node_rack[N]  */
  int64_t ctx_off;          /* ctx pool: counter[N][ctx_width] read at start, written at end;
                               -1 = start from an empty Context and do not write it back     */
} kas_scenario_desc;

/* Per-topic outcome (what one generateAssignment call would have returned/thrown). */
typedef struct kas_topic_result {
  int32_t status;           /* KAS_OK or a KAS_FAIL_* / KAS_SKIPPED code                     */
  int32_t fail_partition;   /* partition id for KAS_FAIL_UNASSIGNABLE, else -1               */
  int32_t moved_replicas;   /* sum_p |set(new[p]) \ set(cur[p])|                             */
  int32_t moved_partitions; /* #{p : set(new[p]) != set(cur[p])}                             */
} kas_topic_result;

/* Per-scenario record: fixed 32 bytes, the unit of the multi-GPU all-gather. */
typedef struct kas_scenario_result {
  int32_t  status;          /* status of the first failing topic, or KAS_OK                  */
  int32_t  fail_topic;      /* index of that topic within the scenario, or -1                */
  int32_t  fail_partition;  /* its fail_partition, or -1                                     */
  int32_t  moved_replicas;  /* summed over the scenario's solved topics                      */
  int32_t  moved_partitions;
  int32_t  reserved;
  uint64_t digest;          /* order-independent 64-bit checksum of every emitted
                               (topic, partition row, replica slot, broker id); see
                               kas_digest_cell()                                             */
} kas_scenario_result;

/* Host view of a batch: descriptors and node tables always live in HOST memory (a plan
 * uploads them once); the bulk tables are passed separately per solve so they can be
 * device resident. */
typedef struct kas_batch_desc {
  int32_t n_scenarios;
  int32_t n_topics;
  const kas_scenario_desc* scenarios;   /* [n_scenarios]                                     */
  const kas_topic_desc*    topics;      /* [n_topics]                                        */
  const int32_t* node_id;               /* node pool, host                                   */
  const int32_t* node_rack;             /* rack pool, host (same offsets as node_id)         */
  int64_t node_pool_len;                /* elements in each node pool                        */
} kas_batch_desc;

/* Bulk tables of one solve.  All pointers are DEVICE pointers for kas_solve_device and
 * HOST pointers for kas_solve_host.  aux/ctx may be NULL when no descriptor refers to them. */
typedef struct kas_tables {
  const int32_t* cur;                   /* cur pool                                          */
  int32_t*       out;                   /* out pool                                          */
  const int32_t* aux;                   /* cur_len / in_partitions / part_id arrays          */
  int32_t*       ctx;                   /* Context counters, in/out                          */
  kas_topic_result*    topic_results;   /* [n_topics]                                        */
  kas_scenario_result* scenario_results;/* [n_scenarios]                                     */
  int64_t cur_len, out_len, aux_len, ctx_len; /* pool sizes in elements (host path copies
                                                 exactly this much; device path ignores)      */
} kas_tables;

typedef struct kas_ctx  kas_ctx;    /* device context: HIP device + stream + scratch         */
typedef struct kas_plan kas_plan;   /* validated batch shape: descriptors + node tables in HBM,
                                       launch geometry, LDS carve-up                          */

/* Contribution of one emitted cell to kas_scenario_result.digest (sum modulo 2^64 over all
 * cells).  Pure function, usable by any checker.  topic = index within the scenario,
 * row = partition row index, slot = position in the preference list. */
#ifndef KAS_ABI_FN
#define KAS_ABI_FN static inline   /* the HIP translation unit adds __host__ __device__ */
#endif
KAS_ABI_FN uint64_t kas_digest_cell(uint32_t topic, uint32_t row, uint32_t slot,
                                       int32_t broker) {
  /* one 32x32->64 multiply-add per cell: the order kernel evaluates this inside its commit path */
  const uint32_t tlo = (topic + 1u) * 0x9E3779B1u, thi = (topic + 1u) * 0x85EBCA77u;
  const uint32_t b = (uint32_t)broker ^ tlo ^ 0x7F4A7C15u;
  const uint32_t r = (((row << 4) | ((slot & 7u) << 1) | 1u)) ^ (thi & 0xFFFFFFFEu);   /* odd */
  uint64_t x = (uint64_t)b * (uint64_t)r + (((uint64_t)r << 32) | (uint64_t)b);
  x ^= x >> 29;
  return x;
}

int         kas_abi_version(void);
const char* kas_strerror(int code);          /* text for KAS_E_* return codes               */
const char* kas_status_string(int status);   /* text for KAS_OK / KAS_FAIL_* status codes   */
const char* kas_last_error(void);            /* thread-local detail of the last failure     */

/* KTA:47-69 as data (host arithmetic, no device): the replication factor generateAssignment resolves from the current
 * assignment when desiredReplicationFactor < 0, and its three precondition checks.  partition_ids[i] / list_sizes[i]: the
 * entries of currentAssignment IN THE ORDER THE CALLER'S MAP ITERATES (KTA:50: the first entry fixes the factor, KTA:57-60: the
 * first later entry of another size fails the topic).  Returns 0 and fills *res: status KAS_OK (rf = the factor to solve with),
 * KAS_FAIL_RF_MISMATCH (fail_partition, fail_list_size = the entry of KTA:58-60; rf = the factor it was held against),
 * KAS_FAIL_RF_NOT_POSITIVE (KTA:65-66) or KAS_FAIL_RF_GT_BROKERS (KTA:67-69; rf = the offending factor). */
typedef struct kas_rf_result {
  int32_t status;
  int32_t rf;
  int32_t fail_partition;   /* -1 unless KAS_FAIL_RF_MISMATCH */
  int32_t fail_list_size;   /* -1 unless KAS_FAIL_RF_MISMATCH */
} kas_rf_result;
int kas_resolve_replication_factor(const int32_t* partition_ids, const int32_t* list_sizes, int32_t n_partitions,
                                   int32_t desired_rf, int32_t n_brokers, kas_rf_result* res);

/* The message of the exception the reference throws for a topic status, character for character (what a binding puts into
 * its IllegalStateException): KAS:183-184 "Partition <p> could not be fully assigned!"; KTA:58-60 "Topic <t> has partition
 * <p> with unexpected replication factor <n>"; KTA:65-66 "Topic <t> does not have a positive replication factor!"; KTA:67-69
 * "Topic <t> has a higher replication factor (<rf>) than available brokers!".  `topic` is UTF-8.  `rf` and `list_size` are
 * read only by the statuses that print them.  Returns the length written (truncated to n - 1, always terminated), 0 for KAS_OK
 * and for statuses the reference has no message for (KAS_FAIL_HASH_INDEX is an ArrayIndexOutOfBoundsException without one). */
int kas_failure_text(const char* topic, int32_t status, int32_t fail_partition, int32_t rf, int32_t list_size, char* buf, int n);

/* Number of visible HIP devices (0 when there is none; never fails). */
int kas_device_count(void);

/* Create / destroy a device context on HIP device `device` (gfx950 required). */
int  kas_ctx_create(int device, kas_ctx** out_ctx);
void kas_ctx_destroy(kas_ctx* ctx);

/* Validate a batch shape, upload descriptors + node tables, size scratch and LDS. */
int  kas_plan_create(kas_ctx* ctx, const kas_batch_desc* batch, kas_plan** out_plan);
void kas_plan_destroy(kas_plan* plan);

/* What a solve of this plan launches, for measurement records: the instantiated kernels with
 * their template arguments (list width class W, wavefronts per scenario workgroup NW, scenarios
 * per solver wavefront G, packed counters), the form of each phase and the launch geometry, as
 * one line of text, e.g.
 *   "kas_fill_kernel<3,4>[quota] grid=1000x256 lds=32016 + kas_order_permutation_kernel +
 *    kas_order_ticket_kernel<3,2,true> grid=500x192 lds=25744"
 * Reflects the current kas_plan_set_flags state.  Returns the length written (excluding the
 * terminating NUL; truncated to n-1), or a negative KAS_E_* code. */
int kas_plan_describe(const kas_plan* plan, char* buf, int n);

/* Bytes of HBM the path must move per solve of this plan, in the plan's own cell width:
 * kas_plan_create (int32 broker ids in and out — SURVEY 8(d)'s yardstick):
 *   sum over topics 4*P*(cur_width + out_width) + sum over scenarios 8*N (+ ctx in/out);
 * kas_plan_create16 (uint16 node indices; no id table is read):
 *   sum over topics 2*P*(cur_width + out_width) + sum over scenarios 4*N (+ ctx in/out). */
int64_t kas_plan_algorithmic_bytes(const kas_plan* plan);

/* Solve with every bulk table already resident in HBM.  `hip_stream` is a hipStream_t
 * (NULL = the context's own stream).  Asynchronous: returns after enqueueing.
 * A plan owns the scratch of ONE solve (orphan lists, accept masks, scenario order, device
 * counters): it supports one solve in flight.  Solves of the same plan are therefore ordered —
 * a solve enqueued on a different stream than the plan's previous one first waits (on the device,
 * hipStreamWaitEvent) for that previous solve to finish.  Callers that want several batches in
 * flight create one plan per batch in flight (what bench.py does). */
int kas_solve_device(kas_plan* plan, const kas_tables* device_tables, void* hip_stream);

/* Block until everything enqueued on the context's own stream has finished. */
int kas_ctx_synchronize(kas_ctx* ctx);

/* Host callers (JNI / ctypes / C++ mirror / the CLI's per-topic loop, KAG:172-186): tables in HOST
 * memory in, results in HOST memory out, blocking.
 * The context keeps what repeated calls can share: its device buffers only ever grow, and the plans
 * of the most recent batch shapes are reused — a plan whose descriptors + node tables are equal byte
 * for byte is used as is, any other one is rebuilt in place over the scratch it already owns (a
 * what-if caller changes the broker sets on every call) — so a caller pays no hipMalloc / hipFree
 * after the first call of a shape.  Calls on one context are serialised.
 * Copies: a pool in memory HIP knows as pinned (kas_host_alloc, hipHostMalloc, hipHostRegister) is
 * moved by DMA straight from / to the caller's buffer; pageable memory goes through the runtime's
 * staging.  Only [first, last] element a descriptor refers to is moved in either direction.  Batches
 * whose tables are large and laid out scenario by scenario are cut into (up to three) scenario ranges, and the
 * upload of one range, the solve of the previous one and the download of the one before run
 * concurrently, on streams of their own.  On an error after work was enqueued the call drains its streams
 * before it returns (no kernel or copy is left touching the caller's memory). */
int kas_solve_host(kas_ctx* ctx, const kas_batch_desc* batch, const kas_tables* host_tables);

/* The what-if form of the same call (KAG:131-187: one snapshot, many broker sets, ONE assignment
 * printed): every scenario is solved and reports its 16-byte topic records and its 32-byte scenario
 * record, but out rows come back only for the scenarios listed in select[0..n_select), packed back to
 * back in that order (per selected scenario its topics in order, P x out_width ints each;
 * host_tables->out_len >= their sum; out may be NULL when n_select == 0).  With every scenario's
 * cur_off pointing at the same rows (the layout of kafka-assigner_amd/whatif.py) a call moves one cur
 * table and the node tables up and a few kilobytes down.  n_select < 0: every row, in place, exactly as
 * kas_solve_host. */
int kas_solve_host_select(kas_ctx* ctx, const kas_batch_desc* batch, const kas_tables* host_tables,
                          const int32_t* select, int32_t n_select);

/* ---- ABI v5: the same host call with 16-bit cells -------------------------------------------------------
 * The host boundary of the per-topic call (KafkaTopicAssigner.java:70-71) and of the CLI's loop over topics
 * (KafkaAssignmentGenerator.java:173-184) is bound by the host link, not by the solve; kas_solve_host16 moves half the
 * bytes.  cur / out cells are uint16 NODE INDICES: cell value i names node i of the scenario's node table — the i-th
 * smallest broker id of its broker set; every caller sorts the ids anyway to lay out node_rack[] (the reference's own
 * TreeSet at KafkaAssignmentStrategy.java:73-99), and maps cells back with one table lookup.  KAS_CELL16_NONE in cur =
 * a broker that is not in the scenario's broker set (its replica is dropped whichever broker it was, KAS:108-110), in
 * out = the pad value (-1 of the int32 layout).  Everything the algorithm derives from broker ids is their ORDER
 * (KAS:73-99 sorted nodes, KAS:188-200 processing order by position, KAS:263-278 ties by position), which the indices
 * keep: rows, statuses, fail_partition and movement counts are those of kas_solve_host on the int32 form of the same
 * batch, cell for cell after the lookup (tests/test_cells16.py).  The digest of a scenario record covers the cells as
 * emitted, i.e. kas_digest_cell over node indices.
 *   batch->node_id is not read and may be NULL (node i has id i); node_rack[], descriptors, aux, ctx, the result
 *   records and select / n_select (n_select < 0: every row in place) are exactly kas_solve_host_select's; descriptor
 *   offsets and cur_len / out_len count cells.  More than 32,767 brokers in a scenario (KAS_N_LIMIT, the limit of every plan: bit 15 of a cell means "no holder" inside the kernels): KAS_E_UNSUPPORTED.
 * On the device the batch is solved on the 16-bit cells themselves where the kernels with that I/O take it
 * (kas_plan_create16 below); any other batch is widened before and narrowed behind an int32 solve (two streaming kernels on
 * the range's solve stream). */
#define KAS_CELL16_NONE 0xFFFFu
typedef struct kas_tables16 {
  const uint16_t* cur;                  /* cur pool, node indices                            */
  uint16_t*       out;                  /* out pool, node indices                            */
  const int32_t* aux;
  int32_t*       ctx;
  kas_topic_result*    topic_results;
  kas_scenario_result* scenario_results;
  int64_t cur_len, out_len, aux_len, ctx_len;
} kas_tables16;
int kas_solve_host16(kas_ctx* ctx, const kas_batch_desc* batch, const kas_tables16* host_tables,
                     const int32_t* select, int32_t n_select);

/* The device-resident form of the same layout: a plan whose solves read and write 16-bit cells in HBM.  The fill kernel
 * then streams 2 bytes a cell instead of 4, the order kernel stores a final row's node indices as they are (no gather of
 * broker ids) over the mid row the row was made from, so a topic's out region is all the scratch its rows need:
 * 6 + 6 instead of 12 + 12 bytes of table traffic per row of three replicas.  Served by the kernels of lists up to 3 wide
 * (fill kernel, kas_p4_kernel, relaxation and round forms of the order kernel): any other batch — lists 4 and more wide —
 * is KAS_E_UNSUPPORTED here, and kas_solve_host16 widens such a batch on the device instead.  batch->node_id is not read; tables as kas_solve_device's, cells as kas_solve_host16's;
 * kas_plan_set_flags: no ticket form (KAS_PLAN_TICKET_ORDER takes the round form); KAS_PLAN_VERIFY_SAMPLE works as on int32 cells (round 6). */
int kas_plan_create16(kas_ctx* ctx, const kas_batch_desc* batch, kas_plan** out_plan);
int kas_solve_device16(kas_plan* plan, const kas_tables16* device_tables, void* hip_stream);

/* Pinned host memory for table pools (DMA without staging: see kas_solve_host).  A JNI caller wraps
 * it with NewDirectByteBuffer, a Python caller with numpy.frombuffer. */
int  kas_host_alloc(int64_t bytes, void** out_ptr);
void kas_host_free(void* ptr);

/* Sharding below Python (mirror of kafka-assigner_amd/sharding.py shard_range): scenarios
 * [*lo, *hi) of `total` belong to rank `rank` of `world` — contiguous, sizes differing by at most one,
 * the larger shards first. */
void kas_shard_range(int64_t total, int32_t rank, int32_t world, int64_t* lo, int64_t* hi);

/* Scenarios [lo, hi) of a batch as a batch of its own: `out` borrows the arrays of `b` (topic and node
 * pools narrowed to what the slice refers to) and uses scen_scratch[hi - lo] for the rebased scenario
 * descriptors; `tables_out` (optional, with `tables`) is the matching view of the result arrays — the
 * bulk pools stay whole, their offsets are absolute.  Topic descriptors of the slice's scenarios must
 * not be shared with scenarios outside it. */
int kas_batch_slice(const kas_batch_desc* b, int64_t lo, int64_t hi, kas_scenario_desc* scen_scratch,
                    kas_batch_desc* out, const kas_tables* tables, kas_tables* tables_out);

/* One batch over several devices: rank r of n_ctx solves kas_shard_range(S, r, n_ctx) on ctxs[r]
 * (one host thread per context, kas_solve_host on its slice), no communication — the scenarios are
 * independent and the caller's host arrays are the gather.  Contexts may sit on the same device.
 * A shard moves the whole extent of `out` / `ctx` its scenarios refer to, so the extents of consecutive
 * shards must not overlap (pools laid out in scenario order, the usual case); a batch whose extents do
 * overlap is solved on ctxs[0] alone — same results, no sharding. */
int kas_solve_host_sharded(kas_ctx* const* ctxs, int32_t n_ctx, const kas_batch_desc* batch,
                           const kas_tables* host_tables);

/* The hardware property the relaxation form of the preference ordering (lists <= 3 wide) and the fill kernel's quota draw
 * rest on — the LDS serves the lanes of one atomic-with-return instruction in ascending lane order; measured, not
 * documented — as this context's self-tests found it (kas_ctx_create, and again at every kas_plan_create of a batch
 * that may take the form, under LDS load on every CU and under partial EXEC masks): 1 held on every lane-operation
 * checked, 0 VIOLATED (the context uses the ticket forms and the draw without return from then on), -1 the test could
 * not run (same consequence), -2 switched off by the environment (KAS_NO_LANE_ORDER=1: the kill switch).
 * *lane_ops_checked (may be NULL): lane-operations checked so far.  See also KAS_PLAN_VERIFY_SAMPLE. */
int kas_ctx_lds_lane_order(const kas_ctx* ctx, int64_t* lane_ops_checked);

/* Counters of the host path since kas_ctx_create: calls, calls that found their plan in the
 * context's cache byte for byte, device allocations made (any pointer may be NULL). */
int kas_ctx_host_stats(kas_ctx* ctx, int64_t* calls, int64_t* plan_hits, int64_t* device_allocs);

/* Average device time in microseconds of one solve (both kernels) over the launches recorded since
 * the last call (HIP events on the launch stream); resets the accumulator. Returns <0 on
 * error, and *launches = 0 when nothing was recorded. */
int kas_plan_kernel_time_us(kas_plan* plan, double* avg_us, int* launches);

/* The same accumulator split in two: a solve is the fill kernel (P0-P3), first fit (P4: kas_p4_kernel behind the fill
 * where kas_plan_describe names it, inside the fill workgroup elsewhere) — together *fill_us — followed by the order
 * kernel (P5, *order_us) on one stream.  Either call resets the accumulator. */
int kas_plan_phase_times_us(kas_plan* plan, double* fill_us, double* order_us, int* launches);

/* Behaviour switches of a plan (default 0); every combination produces identical results, they
 * exist so that the alternative forms can be tested and timed on the same inputs.
 *   KAS_PLAN_GENERIC_FILL  always run the general multi-sweep sticky fill instead of the
 *                          rack-diverse histogram/quota form
 *   KAS_PLAN_ROUND_ORDER   always run the tile-round preference ordering instead of the relaxation / ticket forms
 *   KAS_PLAN_TICKET_ORDER  lists <= 3 wide: the ticket form of the preference ordering (three
 *                          wavefronts per pair of scenarios) where the relaxation form (one wavefront per scenario,
 *                          kas_order_relax.h) would run; KAS_PLAN_WIDE_COUNTERS and KAS_PLAN_GROUPS(n != 0), which
 *                          only mean something to the ticket form, imply it
 *   KAS_PLAN_WIDE_COUNTERS ticket form with 4 x uint16 counter rows even where three 10-bit counts
 *                          in one uint32 would do
 *   KAS_PLAN_TWO_PASS_HIST rack-diverse fill with one histogram for the whole topic and a separate
 *                          chunk-count pass over cur, instead of per-chunk histograms
 *   KAS_PLAN_SPREAD_FILL   take the spread fill (the row scans of large single-topic scenarios over many
 *                          one-wavefront workgroups; chosen by itself for batches of <= 64 scenarios
 *                          of >= 131,072 partitions) for any single-topic batch, with few chunks
 *   KAS_PLAN_RELAX_TILES(n) relaxation form: 1 = tiles of 64 rows, 2 = double tiles (128 rows, two rows per lane: fewer
 *                          LDS round trips per scenario, more LDS operations per row), 0 = by batch size (double
 *                          tiles for batches of fewer than 512 scenarios, where the GPU is not full of wavefronts).
 *                          Lists 4 and 5 wide (round 6): 1 = the relaxation form for these widths (kas_order_relax_wide.h: one
 *                          wavefront per scenario, uint64 counter words; batches without a Context) instead of the wide ticket
 *                          form — exact, and measured slower at BASELINE configs[4], hence opt-in
 *                          3 (round 6) = quad tiles: 256 rows a step, four rows per lane — instances exist on dword mid rows only
 *                          (elsewhere: double tiles); exact, and measured SLOWER than double tiles for a batch alone (order kernel
 *                          1.54 against 1.42 ms per 1000 scenarios), hence only on request
 *   KAS_PLAN_INDEX_ROWS / KAS_PLAN_NO_INDEX_ROWS  rack-diverse fill with per-chunk histograms on int32 cells (round 6): with index
 *                          rows the first row scan — which looks every broker id up, KAS:118-119's nodeMap.get — leaves the row's
 *                          node indices where its mid row goes, the second scan streams those 2-byte cells and stores only the rows
 *                          that do not keep all their replicas: `cur` is read ONCE and every id is looked up once; without, `cur` is
 *                          read by both scans.  Neither flag: the library's default (DESIGN.md section 4.1 has the measurement).
 *   KAS_PLAN_MID32 / KAS_PLAN_NO_MID32  dword mid rows (round 6): between the fill and the order kernel a row of up to three holders
 *                          is ONE aligned dword — the holders sorted by node index, 11 bits each (nothing behind the fill depends on
 *                          their order: first fit appends, KAS:228 sorts) — instead of three uint16 in acceptance order: 4 instead
 *                          of 6 bytes a row written and read, one memory instruction where the packed row takes two, and the order
 *                          kernel's tags become the topic's constants.  Applies to int32 cells, lists 3 wide, at most 2,047 brokers,
 *                          the relaxation form without a Context or sampled verification; otherwise (and with KAS_PLAN_INDEX_ROWS) the
 *                          16-bit rows.  Neither flag: the library's default (DESIGN.md section 4.6 has the measurement).
 *   KAS_PLAN_FULL_FILL     kas_fill_kernel for every scenario.  Default (round 6; int32 cells, lists up to 3 wide, per-chunk
 *                          histograms, a direct id table, first fit handed over): kas_fill_slim_kernel first — the one path
 *                          rack-diverse scenarios take, compiled without the others (120 VGPRs, no scratch) — and
 *                          kas_fill_kernel behind it, on a small grid, for the scenarios it hands back (rows not rack-diverse, ...)
 *   KAS_PLAN_NO_RTN_QUOTA  rack-diverse fill with per-chunk histograms: draw a node's quota with separate LDS atomics,
 *                          reads and a ranking of the tiles in which it runs out, instead of one atomic-with-return
 *                          per list position
 *   KAS_PLAN_FILL_WITH_P4 / KAS_PLAN_SPLIT_P4  first fit (KAS:162-186) inside the fill kernel's workgroup (four wavefronts
 *                          hand the windows over through LDS) / in kas_p4_kernel behind it (one wavefront per scenario),
 *                          whatever the batch size.  Default: kas_p4_kernel for batches of >= 512 scenarios — it
 *                          frees the fill workgroup's registers and LDS early, which is what counts when several
 *                          batches share the GPU — and inside the fill workgroup below that and in kas_solve_host's
 *                          plans (a batch alone on the GPU: the shorter critical path counts).
 *   KAS_PLAN_P4_WITH_ORDER (= both of the above; round 6)  first fit as a second wavefront of the ORDER kernel's workgroup
 *                          (kas_p4_order_kernel; lists up to 3 wide on the relaxation form, no Context, no sampled verification —
 *                          otherwise kas_p4_kernel): the order wavefront follows first fit's progress row by row instead of
 *                          waiting behind a kernel boundary, so a batch that has the GPU to itself lasts
 *                          fill + max(first fit, order) instead of fill + first fit + order
 *   KAS_PLAN_VERIFY_SAMPLE(k) relaxation form: k tiles of 64 rows per topic (evenly spaced, 1..255) are evaluated a second
 *                          time one row at a time — independent of how the LDS orders the lanes of an instruction —
 *                          and a scenario in which a row comes out differently reports KAS_FAIL_WATCHDOG instead of a
 *                          list (an opt-in canary for production: ~3 us per verified tile; the kernel always checks
 *                          that the counters it leaves sum to the rows it retired)
 *   KAS_PLAN_WAVES(n)      wavefronts per scenario workgroup of the fill kernel: 1, 2 or 4
 *   KAS_PLAN_GROUPS(n)     scenarios per wavefront of the ticket-form order kernel: 1, 2 or 4
 *                          (0 = the plan's choice for either) */
#define KAS_PLAN_GENERIC_FILL 1u
#define KAS_PLAN_ROUND_ORDER  2u
#define KAS_PLAN_WIDE_COUNTERS 4u
#define KAS_PLAN_TWO_PASS_HIST 8u
#define KAS_PLAN_FULL_FILL    16u
#define KAS_PLAN_SPREAD_FILL  32u
#define KAS_PLAN_NO_INDEX_ROWS 64u
#define KAS_PLAN_INDEX_ROWS  128u
#define KAS_PLAN_MID32        0x80000u
#define KAS_PLAN_NO_MID32     0x100000u
#define KAS_PLAN_TICKET_ORDER 0x10000u
#define KAS_PLAN_RELAX_TILES(n) (((uint32_t)(n) & 3u) << 17)
#define KAS_PLAN_NO_RTN_QUOTA 0x200000u
#define KAS_PLAN_SPLIT_P4     0x400000u
#define KAS_PLAN_FILL_WITH_P4 0x800000u
#define KAS_PLAN_VERIFY_SAMPLE(k) (((uint32_t)(k) & 0xffu) << 24)
#define KAS_PLAN_WAVES(n)     (((uint32_t)(n) & 0xfu) << 8)
#define KAS_PLAN_GROUPS(n)    (((uint32_t)(n) & 0xfu) << 12)
int kas_plan_set_flags(kas_plan* plan, uint32_t flags);

/* Per-scenario device counters of the plan's most recent solve (after it completed), 16 int64
 * per scenario; times in 10 ns ticks of the constant 100 MHz device clock:
 *   [0] setup  [1] P2 histogram + quota  [2] P2 keep-scan + P3  [3] P4 first fit
 *   [4] P4 windows  [5] P4 node steps  [6] P5 rounds (round form)
 *   [7] P2 tiles of wave 0 that needed quota ranking  [8] order kernel time
 *   ([4], [5], [7], [15] are zero unless the library was built with -DKAS_FILL_COUNTERS)
 *   ticket form: [9] solver steps  [11] of those with rows in hand but none ready
 *                [6] steps that took the queue path  [10] wins / prefix-sum rounds spent there
 *                [14] rows decided inside queues  [15] sum over steps of rows in hand
 *                [12] stager iterations  [13] of those without work
 *   wide ticket form (lists 4, 5 wide): [9]..[11], [6], [14] are the class-1 solver's, [15] = steps of
 *                the first class-0 solver
 *   ([4], [5] count wave 0's share of the P4 windows / node positions)
 * n = capacity of out in int64 elements (>= KAS_STATS_PER_SCENARIO * n_scenarios).  Blocks until the
 * plan's last launch has finished. */
#define KAS_STATS_PER_SCENARIO 16
int kas_plan_stats(kas_plan* plan, int64_t* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* KAS_ABI_H */
