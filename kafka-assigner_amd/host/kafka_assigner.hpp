// kafka_assigner.hpp — C++ host mirror of the reference's operator interface for the hot path,
// over the C ABI of include/kas_abi.h (libkas_hip.so).  Header-only.
//
//   KafkaTopicAssigner.generateAssignment            KafkaTopicAssigner.java:42-72
//   KafkaAssignmentStrategy.getRackAwareAssignment   KafkaAssignmentStrategy.java:40-63
//
// Same names, argument meaning and error behaviour as the reference: Guava Preconditions
// failures surface as kas::IllegalStateException with the reference's message text.  The solve
// itself runs on the GPU; there is no CPU path behind this header.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "kas_abi.h"

namespace kas {

struct IllegalStateException : std::runtime_error {
  explicit IllegalStateException(const std::string& m) : std::runtime_error(m) {}
};
struct ArrayIndexOutOfBoundsException : std::runtime_error {   // KAS:190-192, hashCode() == MIN_VALUE
  explicit ArrayIndexOutOfBoundsException(const std::string& m) : std::runtime_error(m) {}
};
struct SolverError : std::runtime_error {                       // KAS_E_* from the library
  explicit SolverError(const std::string& m) : std::runtime_error(m) {}
};

// java.lang.String.hashCode() over UTF-16 code units of a UTF-8 string (KAS:190).
inline int32_t javaStringHashCode(const std::string& utf8) {
  uint32_t h = 0;
  size_t i = 0;
  auto unit = [&](uint32_t u) { h = 31u * h + u; };
  while (i < utf8.size()) {
    uint32_t c = (unsigned char)utf8[i];
    uint32_t cp;
    int extra;
    if (c < 0x80) { cp = c; extra = 0; }
    else if ((c >> 5) == 0x6) { cp = c & 0x1f; extra = 1; }
    else if ((c >> 4) == 0xe) { cp = c & 0x0f; extra = 2; }
    else { cp = c & 0x07; extra = 3; }
    ++i;
    for (int k = 0; k < extra && i < utf8.size(); ++k, ++i) cp = (cp << 6) | ((unsigned char)utf8[i] & 0x3f);
    if (cp >= 0x10000) {                     // surrogate pair
      cp -= 0x10000;
      unit(0xD800 + (cp >> 10));
      unit(0xDC00 + (cp & 0x3ff));
    } else {
      unit(cp);
    }
  }
  return (int32_t)h;
}

// KafkaAssignmentStrategy.Context (KAS:360-369): leader/follower counters per broker.
struct Context {
  std::map<int, std::map<int, int>> counter;
};

// One process-wide device context (a HIP stream on device 0), created on first use.
inline kas_ctx* deviceContext() {
  static kas_ctx* ctx = nullptr;
  if (!ctx) {
    int rc = kas_ctx_create(0, &ctx);
    if (rc != KAS_E_OK) throw SolverError(std::string("kas_ctx_create: ") + kas_last_error());
  }
  return ctx;
}

class KafkaAssignmentStrategy {
 public:
  // KAS:40-63.  `context` may be null (KAS:59-61).
  static std::map<int, std::vector<int>> getRackAwareAssignment(
      const std::string& topicName, const std::map<int, std::vector<int>>& currentAssignment,
      const std::map<int, std::string>& nodeRackAssignment, const std::set<int>& nodes,
      const std::set<int>& partitions, int replicationFactor, Context* context) {
    const int32_t N = (int32_t)nodes.size();
    std::vector<int32_t> node_id(nodes.begin(), nodes.end());           // ascending (KAS:78)
    std::vector<int32_t> node_rack(N);
    {
      std::map<std::string, int32_t> rack_index;                        // KAS:81-94
      for (int32_t i = 0; i < N; ++i) {
        auto it = nodeRackAssignment.find(node_id[i]);
        const std::string r = it != nodeRackAssignment.end() ? it->second : std::to_string(node_id[i]);
        auto ins = rack_index.emplace(r, (int32_t)rack_index.size());
        node_rack[i] = ins.first->second;
      }
    }
    // rows: keys(currentAssignment) ∪ partitions, ascending (KAS:107-110, 149-150)
    std::set<int> row_ids;
    for (auto& e : currentAssignment) row_ids.insert(e.first);
    for (int p : partitions) row_ids.insert(p);
    const int32_t P = (int32_t)row_ids.size();
    int32_t cw = 0;
    for (auto& e : currentAssignment) cw = std::max(cw, (int32_t)e.second.size());
    const int32_t ow = std::max(std::max(cw, replicationFactor), 1);
    if (ow > KAS_MAX_WIDTH) throw SolverError("replica lists longer than KAS_MAX_WIDTH");
    // 16-bit cells (kas_solve_host16, ABI v5): a replica travels as the position of its broker in node_id[] — half the bytes
    // of the per-topic call (KTA:70-71) over the host link; KAS_CELLS32=1 in the environment keeps int32 broker ids
    const bool cells16 = N <= 32767 && !(getenv("KAS_CELLS32") && getenv("KAS_CELLS32")[0] == '1');
    std::vector<int32_t> aux(3 * (size_t)P), cur(cells16 ? 0 : (size_t)P * std::max(cw, 1), -1);
    std::vector<uint16_t> cur16(cells16 ? (size_t)P * std::max(cw, 1) : 0, (uint16_t)KAS_CELL16_NONE);
    {
      int32_t row = 0;
      for (int p : row_ids) {
        auto it = currentAssignment.find(p);
        const int32_t len = it != currentAssignment.end() ? (int32_t)it->second.size() : 0;
        aux[row] = p;                                                   // part_id
        aux[(size_t)P + row] = len;                                     // cur_len
        aux[2 * (size_t)P + row] = partitions.count(p) ? 1 : 0;         // in_partitions
        for (int32_t k = 0; k < len; ++k) {
          if (!cells16) { cur[(size_t)row * cw + k] = it->second[k]; continue; }
          auto at = std::lower_bound(node_id.begin(), node_id.end(), (int32_t)it->second[k]);
          if (at != node_id.end() && *at == it->second[k]) cur16[(size_t)row * cw + k] = (uint16_t)(at - node_id.begin());
        }
        ++row;
      }
    }
    std::vector<int32_t> ctx((size_t)N * KAS_MAX_WIDTH, 0);
    if (context) {
      for (int32_t i = 0; i < N; ++i) {
        auto it = context->counter.find(node_id[i]);
        if (it == context->counter.end()) continue;
        for (auto& c : it->second)
          if (c.first >= 0 && c.first < KAS_MAX_WIDTH) ctx[(size_t)i * KAS_MAX_WIDTH + c.first] = c.second;
      }
    }
    std::vector<int32_t> out(cells16 ? 0 : (size_t)P * ow + 1, -1);
    std::vector<uint16_t> out16(cells16 ? (size_t)P * ow + 1 : 0, (uint16_t)KAS_CELL16_NONE);

    kas_topic_desc td{};
    td.name_hash = javaStringHashCode(topicName);
    td.n_partitions = P; td.cur_width = cw; td.rf = replicationFactor; td.out_width = ow;
    td.cur_off = 0; td.out_off = 0;
    td.part_id_off = 0; td.cur_len_off = P; td.in_partitions_off = 2 * (int64_t)P;
    kas_scenario_desc sd{};
    sd.n_nodes = N; sd.topic_begin = 0; sd.topic_count = 1;
    sd.ctx_width = context ? KAS_MAX_WIDTH : 0; sd.node_off = 0; sd.ctx_off = context ? 0 : -1;
    kas_batch_desc bd{};
    bd.n_scenarios = 1; bd.n_topics = 1; bd.scenarios = &sd; bd.topics = &td;
    bd.node_id = node_id.data(); bd.node_rack = node_rack.data(); bd.node_pool_len = N;
    kas_topic_result tr{};
    kas_scenario_result sr{};
    kas_tables t{};
    t.cur = cur.data(); t.out = out.data(); t.aux = aux.data(); t.ctx = ctx.data();
    t.topic_results = &tr; t.scenario_results = &sr;
    t.cur_len = (int64_t)P * cw; t.out_len = (int64_t)P * ow; t.aux_len = 3 * (int64_t)P;
    t.ctx_len = context ? (int64_t)N * KAS_MAX_WIDTH : 0;
    int rc;
    if (cells16) {
      kas_tables16 t16{};
      t16.cur = cur16.data(); t16.out = out16.data(); t16.aux = t.aux; t16.ctx = t.ctx;
      t16.topic_results = &tr; t16.scenario_results = &sr;
      t16.cur_len = t.cur_len; t16.out_len = t.out_len; t16.aux_len = t.aux_len; t16.ctx_len = t.ctx_len;
      rc = kas_solve_host16(deviceContext(), &bd, &t16, nullptr, -1);
    } else {
      rc = kas_solve_host(deviceContext(), &bd, &t);
    }
    if (rc != KAS_E_OK) throw SolverError(std::string(kas_strerror(rc)) + ": " + kas_last_error());

    switch (tr.status) {
      case KAS_OK: break;
      case KAS_FAIL_UNASSIGNABLE:                                       // KAS:183-184
        throw IllegalStateException("Partition " + std::to_string(tr.fail_partition) +
                                    " could not be fully assigned!");
      case KAS_FAIL_HASH_INDEX:                                         // KAS:190-192
        throw ArrayIndexOutOfBoundsException("negative node processing index");
      case KAS_FAIL_RF_NOT_POSITIVE:
        throw IllegalStateException("Topic " + topicName + " does not have a positive replication factor!");
      case KAS_FAIL_RF_GT_BROKERS:
        throw IllegalStateException("Topic " + topicName + " has a higher replication factor (" +
                                    std::to_string(replicationFactor) + ") than available brokers!");
      default:
        throw SolverError(std::string("solver status ") + kas_status_string(tr.status));
    }
    std::map<int, std::vector<int>> result;
    {
      int32_t row = 0;
      for (int p : row_ids) {
        std::vector<int> l;
        for (int32_t k = 0; k < ow; ++k) {
          if (cells16) {
            const uint16_t c = out16[(size_t)row * ow + k];
            if (c != KAS_CELL16_NONE) l.push_back(node_id[c]);
          } else {
            const int32_t b = out[(size_t)row * ow + k];
            if (b >= 0) l.push_back(b);
          }
        }
        if (!l.empty()) result[p] = l;                                  // KAS:205-214 lists held rows only
        ++row;
      }
    }
    if (context) {
      for (int32_t i = 0; i < N; ++i) {
        std::map<int, int> c;
        for (int k = 0; k < KAS_MAX_WIDTH; ++k)
          if (ctx[(size_t)i * KAS_MAX_WIDTH + k] != 0) c[k] = ctx[(size_t)i * KAS_MAX_WIDTH + k];
        context->counter[node_id[i]] = c;
      }
    }
    return result;
  }
};

// Mirror of KafkaTopicAssigner (KTA:18-72): one Context per instance (KTA:19-23).
class KafkaTopicAssigner {
 public:
  std::map<int, std::vector<int>> generateAssignment(
      const std::string& topic, const std::map<int, std::vector<int>>& currentAssignment,
      const std::set<int>& brokers, const std::map<int, std::string>& rackAssignment,
      int desiredReplicationFactor) {
    int replicationFactor = desiredReplicationFactor;                   // KTA:49
    std::set<int> partitions;
    for (auto& e : currentAssignment) {                                 // KTA:50-62
      partitions.insert(e.first);
      if (replicationFactor < 0) {
        replicationFactor = (int)e.second.size();
      } else if (desiredReplicationFactor < 0) {
        if (replicationFactor != (int)e.second.size())
          throw IllegalStateException("Topic " + topic + " has partition " + std::to_string(e.first) +
                                      " with unexpected replication factor " + std::to_string(e.second.size()));
      }
    }
    if (!(replicationFactor > 0))                                       // KTA:65-66
      throw IllegalStateException("Topic " + topic + " does not have a positive replication factor!");
    if (!(replicationFactor <= (int)brokers.size()))                    // KTA:67-69
      throw IllegalStateException("Topic " + topic + " has a higher replication factor (" +
                                  std::to_string(replicationFactor) + ") than available brokers!");
    return KafkaAssignmentStrategy::getRackAwareAssignment(topic, currentAssignment, rackAssignment, brokers,
                                                           partitions, replicationFactor, &assignmentContext);
  }

  Context assignmentContext;
};

}  // namespace kas
