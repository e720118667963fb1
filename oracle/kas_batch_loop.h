/*
 * kas_batch_loop.h — batch driver shared by the two CPU solvers under oracle/ (the structure-
 * faithful oracle, kas_oracle.c, and the flat-array baseline, kas_cpu_fast.c).
 * TEST / BENCH INFRASTRUCTURE, NOT PRODUCT: nothing under kafka-assigner_amd/ includes this.
 *
 * Semantics of kas_solve_host (include/kas_abi.h): scenarios are independent; inside a scenario
 * the topics run in order against one Context (KafkaTopicAssigner.java:19-23 keeps one Context
 * per assigner instance) and the first failure skips the rest — the CLI run aborts at the first
 * exception (KafkaAssignmentGenerator.java:173-184).
 *
 * The threaded entry splits the SCENARIOS over n_threads pthreads inside one C call (a shared
 * atomic cursor; every scenario is solved by exactly one thread with its own scratch), which is
 * what "scenario-parallel on all host cores" in BASELINE.md section 3 asks for.
 */
#ifndef KAS_BATCH_LOOP_H
#define KAS_BATCH_LOOP_H

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "kas_abi.h"

/* Per-thread scratch arena.  A scenario needs a few MB of scratch (holder lists, loads, Context
 * counters); calloc / malloc / free per scenario meant an mmap + page faults + munmap each time and,
 * with 256 threads doing that at once, serialisation in the kernel's mm locks: the all-core baselines of
 * round 2 scaled 9-11x on 256 hardware threads.  Each worker thread now owns one block that it bumps
 * through and rewinds (mark / release); a block that turns out too small is replaced by a larger one at
 * the next rewind to empty, so after its first scenario a thread allocates nothing. */
typedef struct kas_arena {
  unsigned char* base;
  size_t cap, used;
  void** spill;           /* allocations that did not fit the block (freed at the rewind to empty); grows as needed */
  int n_spill, spill_cap;
  size_t spill_bytes;
  int depth;              /* marks outstanding: spills are only freed when the outermost one is released */
} kas_arena;

static __thread kas_arena* kas_tls_arena;     /* arena of the scenario loop running on this thread, or NULL */

static void* kas_scratch_alloc(size_t bytes, int zero) {
  kas_arena* a = kas_tls_arena;
  bytes = (bytes + 63) & ~(size_t)63;
  if (bytes == 0) bytes = 64;
  void* p = NULL;
  if (a && a->used + bytes <= a->cap) {
    p = a->base + a->used;
    a->used += bytes;
  } else {
    p = malloc(bytes);
    if (a && p) {
      /* every spilled block is tracked (a scenario of many large topics on a cold arena spills ~6 blocks per
       * topic) and every spilled byte counted, so that the block that replaces the arena is large enough */
      if (a->n_spill == a->spill_cap) {
        const int cap = a->spill_cap ? 2 * a->spill_cap : 64;
        void** grown = (void**)realloc(a->spill, sizeof(void*) * (size_t)cap);
        if (grown) { a->spill = grown; a->spill_cap = cap; }
      }
      if (a->n_spill < a->spill_cap) a->spill[a->n_spill++] = p;
      a->spill_bytes += bytes;
    }
  }
  if (p && zero) memset(p, 0, bytes);
  return p;
}
static size_t kas_scratch_mark(void) {
  if (!kas_tls_arena) return 0;
  kas_tls_arena->depth += 1;
  return kas_tls_arena->used;
}
static void kas_scratch_release(size_t mark) {
  kas_arena* a = kas_tls_arena;
  if (!a) return;
  a->used = mark;
  a->depth -= 1;
  if (a->depth == 0 && a->n_spill > 0) {        /* nothing is live: free what spilled and make room for it next time */
    for (int i = 0; i < a->n_spill; ++i) free(a->spill[i]);
    const size_t want = a->cap + a->spill_bytes + (a->cap + a->spill_bytes) / 4 + 4096;
    free(a->base);
    a->base = (unsigned char*)malloc(want);
    a->cap = a->base ? want : 0;
    a->n_spill = 0; a->spill_bytes = 0;
  }
}
/* Arenas outlive the call that made them: worker threads are created per batch call and a thread sees
 * only S / n_threads scenarios of it (4 at 1000 scenarios on 256 threads), so a block that had to be grown
 * inside every call would never pay for itself.  A thread takes an arena from the process-wide pool when
 * it starts and hands it back, block and all, when it ends. */
#define KAS_ARENA_POOL 1024
static pthread_mutex_t kas_pool_mu = PTHREAD_MUTEX_INITIALIZER;
static kas_arena* kas_pool[KAS_ARENA_POOL];
static int kas_pool_n;

static kas_arena* kas_arena_acquire(void) {
  kas_arena* a = NULL;
  pthread_mutex_lock(&kas_pool_mu);
  if (kas_pool_n > 0) a = kas_pool[--kas_pool_n];
  pthread_mutex_unlock(&kas_pool_mu);
  if (!a) a = (kas_arena*)calloc(1, sizeof(kas_arena));
  kas_tls_arena = a;
  return a;
}
static void kas_arena_return(kas_arena* a) {
  kas_tls_arena = NULL;
  if (!a) return;
  for (int i = 0; i < a->n_spill; ++i) free(a->spill[i]);
  a->n_spill = 0; a->spill_bytes = 0; a->used = 0; a->depth = 0;
  pthread_mutex_lock(&kas_pool_mu);
  if (kas_pool_n < KAS_ARENA_POOL) { kas_pool[kas_pool_n++] = a; a = NULL; }
  pthread_mutex_unlock(&kas_pool_mu);
  if (a) { free(a->base); free(a->spill); free(a); }
}

typedef int (*kas_topic_fn)(int32_t name_hash, int32_t P, const int32_t* part_id,
                            const int32_t* cur, int32_t cur_width, const int32_t* cur_len,
                            const int32_t* in_partitions, int32_t N, const int32_t* node_id,
                            const int32_t* node_rack, int32_t rf, int32_t* counter, int32_t cw,
                            int32_t* out, int32_t out_width, kas_topic_result* res, int64_t* probes);

static void kas_loop_fill_minus_one(int32_t* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = -1;
}

/* one scenario: the per-topic loop of KAG:173-184 over one Context (KAS:360-369) */
static void kas_loop_scenario(const kas_batch_desc* b, const kas_tables* t, int32_t s, kas_topic_fn solve) {
  const kas_scenario_desc* sd = &b->scenarios[s];
  kas_scenario_result* sr = &t->scenario_results[s];
  sr->status = KAS_OK; sr->fail_topic = -1; sr->fail_partition = -1;
  sr->moved_replicas = 0; sr->moved_partitions = 0; sr->reserved = 0; sr->digest = 0;
  const int32_t N = sd->n_nodes;
  const int32_t* node_id = b->node_id + sd->node_off;
  const int32_t* node_rack = b->node_rack + sd->node_off;

  const int32_t cw = KAS_MAX_WIDTH;
  const size_t scratch_mark = kas_scratch_mark();
  int32_t* counter = (int32_t*)kas_scratch_alloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1) * cw, 1);
  if (sd->ctx_off >= 0 && sd->ctx_width > 0)
    for (int32_t n = 0; n < N; ++n)
      for (int32_t r = 0; r < sd->ctx_width && r < cw; ++r)
        counter[(int64_t)n * cw + r] = t->ctx[sd->ctx_off + (int64_t)n * sd->ctx_width + r];

  int failed = 0;
  for (int32_t k = 0; k < sd->topic_count; ++k) {
    const int32_t ti = sd->topic_begin + k;
    const kas_topic_desc* td = &b->topics[ti];
    kas_topic_result* tr = &t->topic_results[ti];
    int32_t* out = t->out + td->out_off;
    if (failed) {
      tr->status = KAS_SKIPPED; tr->fail_partition = -1;
      tr->moved_replicas = 0; tr->moved_partitions = 0;
      kas_loop_fill_minus_one(out, (int64_t)td->n_partitions * td->out_width);
      continue;
    }
    solve(td->name_hash, td->n_partitions,
          td->part_id_off >= 0 ? t->aux + td->part_id_off : NULL,
          t->cur + td->cur_off, td->cur_width,
          td->cur_len_off >= 0 ? t->aux + td->cur_len_off : NULL,
          td->in_partitions_off >= 0 ? t->aux + td->in_partitions_off : NULL,
          N, node_id, node_rack, td->rf, counter, cw, out, td->out_width, tr, NULL);
    if (tr->status != KAS_OK) {
      failed = 1;
      sr->status = tr->status; sr->fail_topic = k; sr->fail_partition = tr->fail_partition;
      continue;
    }
    sr->moved_replicas += tr->moved_replicas;
    sr->moved_partitions += tr->moved_partitions;
    /* (summed in a local: two 32-byte records share a cache line, and a read-modify-write of
     * sr->digest per cell made the threads of neighbouring scenarios fight over it — with that, two
     * threads were SLOWER than one, and 256 threads 9x one) */
    uint64_t dg = 0;
    for (int32_t p = 0; p < td->n_partitions; ++p)
      for (int32_t r = 0; r < td->out_width; ++r) {
        const int32_t v = out[(int64_t)p * td->out_width + r];
        if (v != -1) dg += kas_digest_cell((uint32_t)k, (uint32_t)p, (uint32_t)r, v);
      }
    sr->digest += dg;
  }
  if (sd->ctx_off >= 0 && sd->ctx_width > 0)
    for (int32_t n = 0; n < N; ++n)
      for (int32_t r = 0; r < sd->ctx_width && r < cw; ++r)
        t->ctx[sd->ctx_off + (int64_t)n * sd->ctx_width + r] = counter[(int64_t)n * cw + r];
  kas_scratch_release(scratch_mark);
}

typedef struct {
  const kas_batch_desc* b;
  const kas_tables* t;
  kas_topic_fn solve;
  int32_t next;          /* shared cursor, advanced with __atomic_fetch_add */
} kas_loop_shared;

static void* kas_loop_worker(void* arg) {
  kas_loop_shared* sh = (kas_loop_shared*)arg;
  kas_arena* arena = kas_arena_acquire();
  for (;;) {
    const int32_t s = __atomic_fetch_add(&sh->next, 1, __ATOMIC_RELAXED);
    if (s >= sh->b->n_scenarios) break;
    kas_loop_scenario(sh->b, sh->t, s, sh->solve);
  }
  kas_arena_return(arena);
  return NULL;
}

/* hardware threads of this host (what std::thread::hardware_concurrency() reports) */
static int kas_loop_host_threads(void) {
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  return n > 0 ? (int)n : 1;
}

/* n_threads <= 0: every hardware thread.  Returns the number of threads used, or < 0. */
static int kas_loop_batch(const kas_batch_desc* b, const kas_tables* t, kas_topic_fn solve, int n_threads) {
  if (!b || !t || b->n_scenarios < 0 || b->n_topics < 0) return KAS_E_INVALID_ARG;
  if (n_threads <= 0) n_threads = kas_loop_host_threads();
  if (n_threads > b->n_scenarios) n_threads = b->n_scenarios > 0 ? b->n_scenarios : 1;
  if (n_threads == 1) {
    kas_arena* arena = kas_arena_acquire();
    for (int32_t s = 0; s < b->n_scenarios; ++s) kas_loop_scenario(b, t, s, solve);
    kas_arena_return(arena);
    return 1;
  }
  kas_loop_shared sh;
  sh.b = b; sh.t = t; sh.solve = solve; sh.next = 0;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
  if (!th) return KAS_E_NOMEM;
  int started = 0;
  for (int i = 0; i < n_threads; ++i) {
    if (pthread_create(&th[i], NULL, kas_loop_worker, &sh) != 0) break;
    ++started;
  }
  if (started == 0) kas_loop_worker(&sh);          /* no thread could be created: run inline */
  for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
  free(th);
  return started > 0 ? started : 1;
}

/* How much parallelism does this host actually give the process?  n_threads threads each run the same
 * fixed register-only integer loop; returns the wall time in seconds.  (time at 1 thread) x n / (time at
 * n threads) = the cores the threads really got — a container with a CPU quota reports 256 hardware
 * threads and delivers a fraction of them, which bounds any all-core baseline measured in it. */
typedef struct { uint64_t iters; volatile uint64_t sink; } kas_probe_arg;
static __attribute__((unused)) void* kas_probe_worker(void* p) {
  kas_probe_arg* a = (kas_probe_arg*)p;
  uint64_t x = 88172645463325252ull;
  for (uint64_t i = 0; i < a->iters; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; }
  a->sink = x;
  return NULL;
}
static __attribute__((unused)) double kas_loop_parallelism_probe(int n_threads, uint64_t iters) {
  if (n_threads < 1) n_threads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
  kas_probe_arg* args = (kas_probe_arg*)malloc(sizeof(kas_probe_arg) * (size_t)n_threads);
  if (!th || !args) { free(th); free(args); return -1.0; }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  int started = 0;
  for (int i = 0; i < n_threads; ++i) {
    args[i].iters = iters; args[i].sink = 0;
    if (pthread_create(&th[i], NULL, kas_probe_worker, &args[i]) != 0) break;
    ++started;
  }
  for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  free(th); free(args);
  if (started != n_threads) return -1.0;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

#endif /* KAS_BATCH_LOOP_H */
