// kas_cli.cpp — `kafka-assignment-generator` over a cluster snapshot instead of ZooKeeper.
//
// Mirrors KafkaAssignmentGenerator (KafkaAssignmentGenerator.java): same flags (KAG:53-84), same
// three modes (KAG:86-101), same stdout shape ("CURRENT ASSIGNMENT:" / "CURRENT BROKERS:" /
// "NEW ASSIGNMENT:" followed by one JSON document, KAG:103-129, 185-186), same broker-set
// resolution (KAG:137-151, 189-250) and the same per-topic loop against ONE KafkaTopicAssigner
// (KAG:172-184).  Everything ZooKeeper provided comes from --snapshot <file>:
//
//   { "brokers":    [ {"id": 1, "host": "h1", "port": 9092, "rack": "a"}, ... ],   // PRINT_CURRENT_BROKERS shape
//     "partitions": [ {"topic": "t", "partition": 0, "replicas": [1, 2, 3]}, ... ] } // reassignment-JSON shape
//
// The solve goes through kas::KafkaTopicAssigner (kafka_assigner.hpp) -> C ABI -> HIP kernels.
// A failing topic aborts the run after "CURRENT ASSIGNMENT" was printed, like the uncaught
// IllegalStateException of the reference (exit code 1 here instead of a JVM stack trace).
#include <stdio.h>

#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "kafka_assigner.hpp"
#include "mini_json.hpp"

namespace {

struct Broker { int id; std::string host; int port; bool has_rack; std::string rack; };

struct Snapshot {
  std::vector<Broker> brokers;                                          // ZK order = file order
  std::vector<std::string> topics;                                      // first-appearance order
  std::map<std::string, std::map<int, std::vector<int>>> assignment;
};

std::vector<std::string> split(const std::string& s) {                  // Splitter.on(',') (KAG:52)
  std::vector<std::string> out;
  size_t b = 0;
  for (;;) {
    size_t e = s.find(',', b);
    out.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
    if (e == std::string::npos) break;
    b = e + 1;
  }
  return out;
}

Snapshot loadSnapshot(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open snapshot " + path);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string text = ss.str();
  mjson::Ptr root = mjson::Parser(text).parse();
  if (root->kind != mjson::Value::Object) throw std::runtime_error("snapshot: top level must be an object");
  Snapshot s;
  if (const mjson::Ptr* bs = root->find("brokers")) {
    for (auto& b : (*bs)->a) {
      Broker br{};
      const mjson::Ptr* id = b->find("id");
      if (!id) throw std::runtime_error("snapshot: broker without id");
      br.id = (int)(*id)->i;
      if (const mjson::Ptr* h = b->find("host")) br.host = (*h)->s;
      if (const mjson::Ptr* p = b->find("port")) br.port = (int)(*p)->i;
      if (const mjson::Ptr* r = b->find("rack")) {
        if ((*r)->kind == mjson::Value::String) { br.has_rack = true; br.rack = (*r)->s; }
      }
      s.brokers.push_back(br);
    }
  }
  if (const mjson::Ptr* ps = root->find("partitions")) {
    for (auto& p : (*ps)->a) {
      const mjson::Ptr *t = p->find("topic"), *pn = p->find("partition"), *rs = p->find("replicas");
      if (!t || !pn || !rs) throw std::runtime_error("snapshot: partition entry needs topic, partition, replicas");
      if (!s.assignment.count((*t)->s)) s.topics.push_back((*t)->s);
      std::vector<int> reps;
      for (auto& r : (*rs)->a) reps.push_back((int)r->i);
      s.assignment[(*t)->s][(int)(*pn)->i] = reps;
    }
  }
  return s;
}

// {"version":1,"partitions":[{"topic":..,"partition":..,"replicas":[..]},...]} (KAG:49, 169-186)
std::string reassignmentJson(const std::vector<std::string>& topics,
                             const std::map<std::string, std::map<int, std::vector<int>>>& assignment) {
  auto root = mjson::make(mjson::Value::Object);
  root->o.emplace_back("version", mjson::integer(1));
  auto parts = mjson::make(mjson::Value::Array);
  for (auto& t : topics) {
    auto it = assignment.find(t);
    if (it == assignment.end()) continue;
    for (auto& e : it->second) {                                         // ascending partition (TreeMap)
      auto pj = mjson::make(mjson::Value::Object);
      pj->o.emplace_back("topic", mjson::string(t));
      pj->o.emplace_back("partition", mjson::integer(e.first));
      auto reps = mjson::make(mjson::Value::Array);
      for (int b : e.second) reps->a.push_back(mjson::integer(b));
      pj->o.emplace_back("replicas", reps);
      parts->a.push_back(pj);
    }
  }
  root->o.emplace_back("partitions", parts);
  return mjson::dump(root);
}

void usage() {
  fprintf(stderr,
          "./kafka-assignment-generator.sh [options...] arguments...\n"
          " --snapshot FILE                  : cluster snapshot JSON (stands in for --zk_string)\n"
          " --zk_string VAL                  : accepted and ignored when --snapshot is given\n"
          " --mode [PRINT_CURRENT_ASSIGNMENT | PRINT_CURRENT_BROKERS | PRINT_REASSIGNMENT]\n"
          " --integer_broker_ids VAL         : comma-separated list of Kafka broker IDs (integers)\n"
          " --broker_hosts VAL               : comma-separated list of broker hostnames (instead of broker IDs)\n"
          " --broker_hosts_to_remove VAL     : comma-separated list of broker hostnames to exclude\n"
          " --topics VAL                     : comma-separated list of topics\n"
          " --desired_replication_factor N   : change the replication factor (default: keep)\n"
          " --disable_rack_awareness         : ignore rack configurations\n");
}

std::set<int> hostnamesToIds(const Snapshot& s, const std::set<std::string>& hosts, bool checkPresence) {
  std::set<int> ids;                                                     // KAG:189-204
  for (auto& b : s.brokers) if (hosts.count(b.host)) ids.insert(b.id);
  if (checkPresence && hosts.size() != ids.size()) {
    std::string found;
    for (int id : ids) found += (found.empty() ? "" : ", ") + std::to_string(id);
    throw std::invalid_argument("Some hostnames could not be found! We found: [" + found + "]");
  }
  return ids;
}

}  // namespace

int main(int argc, char** argv) {
  std::string snapshot, mode, brokerIds, brokerHosts, hostsToRemove, topicsArg;
  bool haveIds = false, haveHosts = false, haveTopics = false, disableRack = false, haveZk = false;
  int desiredRf = -1;
  bool bad = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&](std::string& dst) { if (i + 1 < argc) dst = argv[++i]; else bad = true; };
    std::string tmp;
    if (a == "--snapshot") val(snapshot);
    else if (a == "--zk_string") { val(tmp); haveZk = true; }
    else if (a == "--mode") val(mode);
    else if (a == "--integer_broker_ids") { val(brokerIds); haveIds = true; }
    else if (a == "--broker_hosts") { val(brokerHosts); haveHosts = true; }
    else if (a == "--broker_hosts_to_remove") val(hostsToRemove);
    else if (a == "--topics") { val(topicsArg); haveTopics = true; }
    else if (a == "--desired_replication_factor") { val(tmp); try { desiredRf = std::stoi(tmp); } catch (...) { bad = true; } }
    else if (a == "--disable_rack_awareness") disableRack = true;
    else bad = true;
  }
  (void)haveZk;
  // KAG:258-270: on any argument problem print the usage and return normally (exit code 0)
  if (bad || snapshot.empty() || mode.empty() || (haveIds && haveHosts) ||
      (mode != "PRINT_CURRENT_ASSIGNMENT" && mode != "PRINT_CURRENT_BROKERS" && mode != "PRINT_REASSIGNMENT")) {
    usage();
    return 0;
  }
  try {
    const Snapshot snap = loadSnapshot(snapshot);
    std::vector<std::string> topics = haveTopics ? split(topicsArg) : snap.topics;   // KAG:155-157

    if (mode == "PRINT_CURRENT_BROKERS") {                               // KAG:113-129
      auto arr = mjson::make(mjson::Value::Array);
      for (auto& b : snap.brokers) {
        auto o = mjson::make(mjson::Value::Object);
        o->o.emplace_back("id", mjson::integer(b.id));
        o->o.emplace_back("host", mjson::string(b.host));
        o->o.emplace_back("port", mjson::integer(b.port));
        if (b.has_rack) o->o.emplace_back("rack", mjson::string(b.rack));
        arr->a.push_back(o);
      }
      std::cout << "CURRENT BROKERS:\n" << mjson::dump(arr) << std::endl;
      return 0;
    }
    if (mode == "PRINT_CURRENT_ASSIGNMENT") {                            // KAG:103-111
      std::cout << "CURRENT ASSIGNMENT:\n" << reassignmentJson(topics, snap.assignment) << std::endl;
      return 0;
    }

    // ---- PRINT_REASSIGNMENT (KAG:131-187) ---------------------------------------------------
    std::set<int> brokerSet;                                             // KAG:206-225
    if (haveIds && !brokerIds.empty()) {
      for (auto& tok : split(brokerIds)) {
        try { size_t used = 0; int v = std::stoi(tok, &used); if (used != tok.size()) throw 1; brokerSet.insert(v); }
        catch (...) { throw std::invalid_argument("Invalid broker ID: " + tok); }
      }
    } else if (haveHosts && !brokerHosts.empty()) {
      auto hs = split(brokerHosts);
      brokerSet = hostnamesToIds(snap, std::set<std::string>(hs.begin(), hs.end()), true);
    }
    std::set<int> excluded;                                              // KAG:227-236
    if (!hostsToRemove.empty()) {
      auto hs = split(hostsToRemove);
      excluded = hostnamesToIds(snap, std::set<std::string>(hs.begin(), hs.end()), false);
    }
    std::map<int, std::string> rackAssignment;                           // KAG:238-250
    if (!disableRack)
      for (auto& b : snap.brokers) if (b.has_rack) rackAssignment[b.id] = b.rack;
    if (brokerSet.empty())                                               // KAG:137-147: all live brokers
      for (auto& b : snap.brokers) brokerSet.insert(b.id);
    std::set<int> brokers;                                               // KAG:150
    for (int b : brokerSet) if (!excluded.count(b)) brokers.insert(b);
    for (auto it = rackAssignment.begin(); it != rackAssignment.end();)  // KAG:151
      it = brokers.count(it->first) ? std::next(it) : rackAssignment.erase(it);

    std::cout << "CURRENT ASSIGNMENT:\n" << reassignmentJson(topics, snap.assignment) << std::endl;   // KAG:160

    std::map<std::string, std::map<int, std::vector<int>>> finalAssignment;
    kas::KafkaTopicAssigner assigner;                                    // ONE Context for the run (KAG:172)
    for (auto& topic : topics) {                                         // KAG:173-184
      auto it = snap.assignment.find(topic);
      static const std::map<int, std::vector<int>> none;
      finalAssignment[topic] = assigner.generateAssignment(topic, it != snap.assignment.end() ? it->second : none,
                                                           brokers, rackAssignment, desiredRf);
    }
    std::cout << "NEW ASSIGNMENT:\n" << reassignmentJson(topics, finalAssignment) << std::endl;       // KAG:185-186
    return 0;
  } catch (const kas::IllegalStateException& e) {
    fprintf(stderr, "Exception in thread \"main\" java.lang.IllegalStateException: %s\n", e.what());
    return 1;
  } catch (const kas::ArrayIndexOutOfBoundsException& e) {
    fprintf(stderr, "Exception in thread \"main\" java.lang.ArrayIndexOutOfBoundsException: %s\n", e.what());
    return 1;
  } catch (const std::invalid_argument& e) {
    fprintf(stderr, "Exception in thread \"main\" java.lang.IllegalArgumentException: %s\n", e.what());
    return 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
}
