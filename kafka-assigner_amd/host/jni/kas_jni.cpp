// kas_jni.cpp — thin JNI shim over the C ABI of include/kas_abi.h.
//
// Replaces the body of KafkaAssignmentStrategy.getRackAwareAssignment
// (KafkaAssignmentStrategy.java:40-63) as called from KafkaTopicAssigner.java:70-71.
// Compiled only where a JDK provides <jni.h>; this repository's image has none, so the
// translation unit is empty there and build() reports it as skipped.
#if defined(__has_include)
#if __has_include(<jni.h>)
#define KAS_HAVE_JNI 1
#endif
#endif

#ifdef KAS_HAVE_JNI
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include <mutex>

#include "kas_abi.h"

namespace {
std::mutex g_mu;          // one kas_ctx (= one HIP stream) for the JVM; callers are serialised
kas_ctx* g_ctx = nullptr;
constexpr int kHeaderInts = 8;
constexpr int kCtxWidth = KAS_MAX_WIDTH;
}  // namespace

extern "C" JNIEXPORT jint JNICALL
Java_siftscience_kafka_tools_NativeAssignmentStrategy_solveBatch(JNIEnv* env, jclass, jobject jin, jobject jout) {
  int32_t* in = static_cast<int32_t*>(env->GetDirectBufferAddress(jin));
  int32_t* out = static_cast<int32_t*>(env->GetDirectBufferAddress(jout));
  if (!in || !out) return KAS_E_INVALID_ARG;
  const int64_t in_ints = env->GetDirectBufferCapacity(jin) / 4, out_ints = env->GetDirectBufferCapacity(jout) / 4;
  const int32_t hash = in[0], P = in[1], cw = in[2], rf = in[3], ow = in[4], N = in[5], has_ctx = in[6];
  if (P < 0 || N < 0 || cw < 0 || ow < 1 || ow > KAS_MAX_WIDTH || cw > ow) return KAS_E_INVALID_ARG;
  const int64_t need_in = kHeaderInts + 2ll * N + 3ll * P + (int64_t)P * cw + (int64_t)N * kCtxWidth;
  const int64_t need_out = 4 + (int64_t)P * ow + (int64_t)N * kCtxWidth;
  if (in_ints < need_in || out_ints < need_out) return KAS_E_INVALID_ARG;

  const int32_t* node_id = in + kHeaderInts;
  const int32_t* node_rack = node_id + N;
  const int32_t* aux = node_rack + N;                 // partId[P], curLen[P], inPartitions[P]
  const int32_t* cur = aux + 3ll * P;
  const int32_t* ctx_in = cur + (int64_t)P * cw;
  int32_t* out_rows = out + 4;
  int32_t* ctx_out = out_rows + (int64_t)P * ow;
  memcpy(ctx_out, ctx_in, sizeof(int32_t) * (size_t)N * kCtxWidth);

  kas_topic_desc td;
  memset(&td, 0, sizeof(td));
  td.name_hash = hash; td.n_partitions = P; td.cur_width = cw; td.rf = rf; td.out_width = ow;
  td.cur_off = 0; td.out_off = 0;
  td.part_id_off = 0; td.cur_len_off = P; td.in_partitions_off = 2ll * P;
  kas_scenario_desc sd;
  memset(&sd, 0, sizeof(sd));
  sd.n_nodes = N; sd.topic_begin = 0; sd.topic_count = 1;
  sd.ctx_width = has_ctx ? kCtxWidth : 0; sd.node_off = 0; sd.ctx_off = has_ctx ? 0 : -1;
  kas_batch_desc bd;
  memset(&bd, 0, sizeof(bd));
  bd.n_scenarios = 1; bd.n_topics = 1; bd.scenarios = &sd; bd.topics = &td;
  bd.node_id = node_id; bd.node_rack = node_rack; bd.node_pool_len = N;

  kas_topic_result tr;
  kas_scenario_result sr;
  kas_tables t;
  memset(&t, 0, sizeof(t));
  t.cur = cur; t.out = out_rows; t.aux = aux; t.ctx = ctx_out;
  t.topic_results = &tr; t.scenario_results = &sr;
  t.cur_len = (int64_t)P * cw; t.out_len = (int64_t)P * ow; t.aux_len = 3ll * P;
  t.ctx_len = has_ctx ? (int64_t)N * kCtxWidth : 0;

  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_ctx) {
    int rc = kas_ctx_create(0, &g_ctx);
    if (rc != KAS_E_OK) return rc;
  }
  int rc = kas_solve_host(g_ctx, &bd, &t);
  if (rc != KAS_E_OK) return rc;
  out[0] = tr.status; out[1] = tr.fail_partition; out[2] = tr.moved_replicas; out[3] = tr.moved_partitions;
  return KAS_E_OK;
}
#endif  // KAS_HAVE_JNI
