// kas_jni.cpp — thin JNI shim over the C ABI of include/kas_abi.h.
//
// Replaces the body of KafkaAssignmentStrategy.getRackAwareAssignment
// (KafkaAssignmentStrategy.java:40-63) as called from KafkaTopicAssigner.java:70-71 — and, because
// the payload is a whole batch (S scenarios x their topics, the kas_batch_desc shape), also gives
// Java callers the batch path: one call per PRINT_REASSIGNMENT run (all topics of
// KafkaAssignmentGenerator.java:172-184 against one Context) or per set of what-if broker sets.
// Compiled only where a JDK provides <jni.h>; this repository's image has none, so the
// translation unit is empty there, build() reports it as skipped, and tests/test_jni_shim.py
// compiles it against tests/jni_stub/jni.h.
//
// Payload (int32 units, native byte order; written by NativeAssignmentStrategy.java):
//   in :  header[12] = {KAS_JNI_LAYOUT, S, T, nodePoolLen, curLen, auxLen, ctxLen, outLen,
//                       device, nSelect, cellBits, 0}
//         scenario descriptors  S x 8 ints   (kas_scenario_desc, 32 bytes each)
//         topic descriptors     T x 16 ints  (kas_topic_desc, 64 bytes each)
//         nodeId[nodePoolLen] nodeRack[nodePoolLen] cur[curLen] aux[auxLen] ctx[ctxLen] select[max(nSelect, 0)]
//   out:  topicResults T x 4 ints, scenarioResults S x 8 ints, out[outLen], ctx[ctxLen]
// Offsets inside the descriptors index these pools exactly as in kas_abi.h.  device = HIP device the
// call runs on (one kas_ctx per device, created on first use).  nSelect = -1: out[] holds every row in
// place (kas_solve_host); nSelect >= 0: the what-if form (kas_solve_host_select) — every scenario
// reports its records, out[] holds only the rows of the scenarios select[] names, packed in that order
// (KAG:177-186 prints ONE assignment however many broker sets were tried).
// Layout 4 (ABI v5): cellBits = 16 -> cur[] and out[] hold uint16 node indices (kas_solve_host16: position of the broker
// in the scenario's ascending node table, 0xFFFF = not in the broker set / pad), two to an int, (curLen + 1) / 2 and
// (outLen + 1) / 2 ints; curLen / outLen and the descriptors' offsets count cells; nodeId[] is carried and not read.
// cellBits = 0 or 32: int32 cells as in layout 3, whose payloads are still accepted.
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
#include <vector>

#include "kas_abi.h"

#define KAS_JNI_LAYOUT 4

namespace {
std::mutex g_mu;          // guards the lazily created contexts; kas_solve_host serialises its own callers
std::vector<kas_ctx*> g_ctx;   // one per HIP device
constexpr int kHeaderInts = 12;
}  // namespace

extern "C" JNIEXPORT jint JNICALL
Java_siftscience_kafka_tools_NativeAssignmentStrategy_solveBatch(JNIEnv* env, jclass, jobject jin, jobject jout) {
  const int32_t* in = static_cast<const int32_t*>(env->GetDirectBufferAddress(jin));
  int32_t* out = static_cast<int32_t*>(env->GetDirectBufferAddress(jout));
  if (!in || !out) return KAS_E_INVALID_ARG;
  const int64_t in_ints = env->GetDirectBufferCapacity(jin) / 4, out_ints = env->GetDirectBufferCapacity(jout) / 4;
  if (in_ints < kHeaderInts || (in[0] != KAS_JNI_LAYOUT && in[0] != 3)) return KAS_E_INVALID_ARG;
  const int64_t S = in[1], T = in[2], npool = in[3], cur_len = in[4], aux_len = in[5], ctx_len = in[6], out_len = in[7];
  const int32_t device = in[8], n_select = in[9];
  const bool cells16 = in[0] >= 4 && in[10] == 16;
  if (in[0] >= 4 && in[10] != 0 && in[10] != 16 && in[10] != 32) return KAS_E_INVALID_ARG;
  const int64_t cur_ints = cells16 ? (cur_len + 1) / 2 : cur_len, out_ints_rows = cells16 ? (out_len + 1) / 2 : out_len;
  if (S < 0 || T < 0 || npool < 0 || cur_len < 0 || aux_len < 0 || ctx_len < 0 || out_len < 0 || device < 0 || n_select < -1)
    return KAS_E_INVALID_ARG;
  const int64_t need_in = kHeaderInts + 8 * S + 16 * T + 2 * npool + cur_ints + aux_len + ctx_len + (n_select > 0 ? n_select : 0);
  const int64_t need_out = 4 * T + 8 * S + out_ints_rows + ctx_len;
  if (in_ints < need_in || out_ints < need_out) return KAS_E_INVALID_ARG;

  // descriptors are copied out of the buffer: no alignment assumption on the ByteBuffer
  static_assert(sizeof(kas_scenario_desc) == 32 && sizeof(kas_topic_desc) == 64, "payload layout");
  std::vector<kas_scenario_desc> scen((size_t)S);
  std::vector<kas_topic_desc> topics((size_t)T);
  const int32_t* p = in + kHeaderInts;
  if (S) memcpy(scen.data(), p, sizeof(kas_scenario_desc) * (size_t)S);
  p += 8 * S;
  if (T) memcpy(topics.data(), p, sizeof(kas_topic_desc) * (size_t)T);
  p += 16 * T;
  const int32_t* node_id = p; p += npool;
  const int32_t* node_rack = p; p += npool;
  const int32_t* cur = p; p += cur_ints;
  const int32_t* aux = p; p += aux_len;
  const int32_t* ctx_in = p; p += ctx_len;
  const int32_t* select = p;

  static_assert(sizeof(kas_topic_result) == 16 && sizeof(kas_scenario_result) == 32, "payload layout");
  std::vector<kas_topic_result> tr((size_t)T);
  std::vector<kas_scenario_result> sr((size_t)S);
  int32_t* out_tr = out;
  int32_t* out_sr = out_tr + 4 * T;
  int32_t* out_rows = out_sr + 8 * S;
  int32_t* ctx_out = out_rows + out_ints_rows;
  if (ctx_len) memcpy(ctx_out, ctx_in, sizeof(int32_t) * (size_t)ctx_len);     // Context counters are in/out

  kas_batch_desc bd;
  memset(&bd, 0, sizeof(bd));
  bd.n_scenarios = (int32_t)S; bd.n_topics = (int32_t)T;
  bd.scenarios = scen.data(); bd.topics = topics.data();
  bd.node_id = node_id; bd.node_rack = node_rack; bd.node_pool_len = npool;
  kas_tables t;
  memset(&t, 0, sizeof(t));
  t.cur = cur; t.out = out_rows; t.aux = aux_len ? aux : nullptr; t.ctx = ctx_len ? ctx_out : nullptr;
  t.topic_results = tr.data(); t.scenario_results = sr.data();
  t.cur_len = cur_len; t.out_len = out_len; t.aux_len = aux_len; t.ctx_len = ctx_len;

  kas_ctx* ctx = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_ctx.empty()) g_ctx.assign((size_t)(kas_device_count() > 0 ? kas_device_count() : 1), nullptr);
    if ((size_t)device >= g_ctx.size()) return KAS_E_INVALID_ARG;
    if (!g_ctx[(size_t)device]) {
      int rc = kas_ctx_create(device, &g_ctx[(size_t)device]);
      if (rc != KAS_E_OK) return rc;
    }
    ctx = g_ctx[(size_t)device];
  }
  // the context keeps its device buffers and the plans of recent batch shapes: a JVM that calls
  // once per topic or per what-if round pays no allocation after the first call
  int rc;
  if (cells16) {
    kas_tables16 t16;
    memset(&t16, 0, sizeof(t16));
    t16.cur = reinterpret_cast<const uint16_t*>(cur); t16.out = reinterpret_cast<uint16_t*>(out_rows);
    t16.aux = t.aux; t16.ctx = t.ctx; t16.topic_results = t.topic_results; t16.scenario_results = t.scenario_results;
    t16.cur_len = cur_len; t16.out_len = out_len; t16.aux_len = aux_len; t16.ctx_len = ctx_len;
    bd.node_id = nullptr;                                      // (node i has id i: the Java side maps the cells back)
    rc = kas_solve_host16(ctx, &bd, &t16, n_select < 0 ? nullptr : select, n_select);
  } else {
    rc = n_select < 0 ? kas_solve_host(ctx, &bd, &t) : kas_solve_host_select(ctx, &bd, &t, select, n_select);
  }
  if (rc != KAS_E_OK) return rc;
  if (T) memcpy(out_tr, tr.data(), sizeof(kas_topic_result) * (size_t)T);
  if (S) memcpy(out_sr, sr.data(), sizeof(kas_scenario_result) * (size_t)S);
  return KAS_E_OK;
}
#endif  // KAS_HAVE_JNI
