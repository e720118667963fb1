#!/usr/bin/env python
"""Compare the reference JVM's answer (tools/JavaGolden.java output) with the MI355X solver's on the
same cluster snapshot, as PARSED JSON (key order of org.json is JVM dependent).

  python tools/compare_golden.py snapshot.json java.json [--disable_rack_awareness]
      [--desired_replication_factor N]

Runs kafka-assigner_amd/host/kafka-assignment-generator --mode PRINT_REASSIGNMENT on the snapshot
(needs the MI355X) with --integer_broker_ids = the snapshot's "solve_brokers" (or all brokers) and
checks: same failure (topic + message) or list-equal replicas for every (topic, partition).
Exit status 0 = identical."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("snapshot")
    ap.add_argument("java_json")
    ap.add_argument("--disable_rack_awareness", action="store_true")
    ap.add_argument("--desired_replication_factor", type=int, default=-1)
    a = ap.parse_args()
    from kafka_assigner_amd import build as kbuild
    cli = kbuild.build_host()
    snap = json.load(open(a.snapshot))
    java = json.load(open(a.java_json))
    ids = snap.get("solve_brokers") or [b["id"] for b in snap["brokers"]]
    cmd = [cli, "--snapshot", a.snapshot, "--mode", "PRINT_REASSIGNMENT",
           "--integer_broker_ids", ",".join(str(b) for b in ids)]
    if a.disable_rack_awareness:
        cmd.append("--disable_rack_awareness")
    if a.desired_replication_factor >= 0:
        cmd += ["--desired_replication_factor", str(a.desired_replication_factor)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if "failed" in java:
        want = java["failed"]["message"]
        ok = r.returncode != 0 and want in r.stderr
        print("reference failed with:", want)
        print("solver:", "same failure" if ok else f"exit {r.returncode}: {r.stderr.strip()[-300:]}")
        return 0 if ok else 1
    if r.returncode != 0:
        print("solver failed where the reference did not:", r.stderr.strip()[-300:])
        return 1
    lines = r.stdout.splitlines()
    mine = json.loads(lines[lines.index("NEW ASSIGNMENT:") + 1])
    key = lambda p: (p["topic"], p["partition"])
    a_map = {key(p): p["replicas"] for p in java["partitions"]}
    b_map = {key(p): p["replicas"] for p in mine["partitions"]}
    bad = [k for k in sorted(set(a_map) | set(b_map)) if a_map.get(k) != b_map.get(k)]
    print(f"{len(a_map)} partitions from the JVM, {len(b_map)} from the solver, {len(bad)} differ")
    for k in bad[:10]:
        print("  ", k, "jvm", a_map.get(k), "solver", b_map.get(k))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
