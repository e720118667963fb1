#!/usr/bin/env python
"""Write seeded scenarios of the bench / parity generator as cluster-snapshot JSON files, one per
scenario, for the JVM-side harness (tools/JavaGolden.java, tools/JavaRefBench.java) and for
`kafka-assignment-generator --snapshot`.  CPU only (numpy); the current assignments are the
generator's G(seed + s, P, N, R, RF), the broker set the scenario's action.

  python tools/export_scenarios.py --out /tmp/scen --scenarios 4 --partitions 10000 --brokers 100 \
      --racks 10 --rf 3 --actions remove1,add_k

Each file: {"brokers": [{"id", "host", "port", "rack"}...]   (old and added brokers),
            "solve_brokers": [ids the scenario solves with],
            "action": "...",
            "partitions": [{"topic": "t0", "partition": p, "replicas": [...]}...]}
`--integer_broker_ids` for the CLI is the comma-joined "solve_brokers".
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kafka_assigner_amd import generator as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--scenarios", type=int, default=4)
    ap.add_argument("--partitions", type=int, default=10000)
    ap.add_argument("--brokers", type=int, default=100)
    ap.add_argument("--racks", type=int, default=10)
    ap.add_argument("--rf", type=int, default=3)
    ap.add_argument("--actions", default=",".join(G.ACTIONS))
    ap.add_argument("--topic", default="t0")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    acts = tuple(a.actions.split(","))
    for s in range(a.scenarios):
        cur = G.random_assignment(a.seed + s, a.partitions, a.brokers, a.racks, a.rf)
        act, bs = G.scenario_action(a.seed, s, a.brokers, a.racks, actions=acts, max_add=max(2, a.brokers // 10))
        ids = sorted(set(range(a.brokers)) | set(int(b) for b in bs.node_id))
        snap = {
            "action": act,
            "brokers": [{"id": b, "host": f"kafka-{b}.example.com", "port": 9092, "rack": "r%02d" % (b % a.racks)}
                        for b in ids],
            "solve_brokers": [int(b) for b in bs.node_id],
            "partitions": [{"topic": a.topic, "partition": p, "replicas": [int(x) for x in cur[p]]}
                           for p in range(a.partitions)],
        }
        path = os.path.join(a.out, f"scen_{s:04d}.json")
        with open(path, "w") as f:
            json.dump(snap, f)
        print(path, act, len(snap["solve_brokers"]), "brokers")


if __name__ == "__main__":
    main()
