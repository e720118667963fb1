#!/usr/bin/env python
"""tools/jvm_check.py — SURVEY.md 8(f) N4 in one command, for a machine that has a JDK (this image has none: nothing here
has ever run this file's Java half).  Driven by `make -C tools jvm-check REF=/path/to/kafka-assigner`:

  1. every vector of tests/golden/survey_appendix_b.json (the four KafkaTopicAssignerTest inputs, the config-1 cases with
     their shared Context, the quirk cases) is written as a cluster snapshot and solved by the UNTOUCHED reference classes
     through tools/JavaGolden.java; the answers are written to tests/golden/survey_appendix_b.jvm.json and diffed against the
     committed `expected` lists (exact list ORDER: the thing this repository can only pin by source reading);
  2. --bench N bench-shaped scenarios (tools/export_scenarios.py: G(seed + s, 100k, 1k, 20, 3) + the bench action mix) are
     timed by tools/JavaRefBench.java and the result is printed in bench.py's cpu_baseline format with kind "reference" —
     BASELINE.md's row B0.

Exit status 0 = every vector identical.  TEST / MEASUREMENT TOOLING: nothing on the product path imports this."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "survey_appendix_b.json")


def snapshot(topics, brokers, racks):
    """topics: [(name, {partition: [replicas]})] in order; every broker that appears anywhere is listed (the reference reads
    racks of live brokers only: KafkaAssignmentGenerator.java:238-250), the solve set goes in --brokers."""
    ids = set(int(b) for b in brokers)
    for _, cur in topics:
        for reps in cur.values():
            ids.update(int(r) for r in reps)
    return {"brokers": [dict({"id": b, "host": "kafka-%d" % b, "port": 9092}, **({"rack": racks[str(b)]} if str(b) in racks else {}))
                        for b in sorted(ids)],
            "partitions": [{"topic": name, "partition": int(p), "replicas": [int(r) for r in reps]}
                           for name, cur in topics for p, reps in sorted(cur.items(), key=lambda kv: int(kv[0]))]}


def java_golden(cp, snap, brokers, desired_rf=-1, norack=False):
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(snap, f)
        path = f.name
    cmd = ["java", "-cp", cp, "siftscience.kafka.tools.JavaGolden", path, "--brokers", ",".join(str(int(b)) for b in brokers)]
    if desired_rf >= 0:
        cmd += ["--desired_replication_factor", str(desired_rf)]
    if norack:
        cmd += ["--disable_rack_awareness"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    os.unlink(path)
    if r.returncode != 0 and not r.stdout.strip():
        # an exception the tool does not catch (the KAS:190 ArrayIndexOutOfBounds is a RuntimeException and is caught)
        return {"crashed": r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "exit %d" % r.returncode}
    return json.loads(r.stdout.strip().splitlines()[-1])


def lists_of(answer, topic):
    return {str(p["partition"]): p["replicas"] for p in answer.get("partitions", []) if p["topic"] == topic}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cp", required=True, help="class path: compiled tools/*.java + the reference's classes + its dependencies")
    ap.add_argument("--bench", type=int, default=8, help="bench-shaped scenarios for JavaRefBench (0: skip)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "survey_appendix_b.jvm.json"))
    a = ap.parse_args()
    fx = json.load(open(FIXTURE))
    regenerated, bad = {"ktat": [], "quirks": [], "config1": []}, []

    def check(name, got, want, what):
        ok = got == want
        print(("ok       " if ok else "DIFFERENT"), what, name)
        if not ok:
            bad.append((what, name, got, want))

    for kind in ("ktat", "quirks"):
        for c in fx[kind]:
            ans = java_golden(a.cp, snapshot([(c["topic"], c["current"])], c["brokers"], c.get("racks", {})), c["brokers"],
                              desired_rf=c.get("desired_rf", -1), norack=not c.get("racks"))
            regenerated[kind].append({"name": c["name"], "answer": ans})
            if "expected" in c:
                check(c["name"], lists_of(ans, c["topic"]), c["expected"], kind)
            else:                                               # Q7: the reference dies of an index error (KAS:190-192)
                check(c["name"], bool(ans.get("failed") or ans.get("crashed")), True, kind + " (must fail)")
    c1 = fx["config1"]
    topics = [(name, cur) for name, cur in zip(c1["topics"], c1["current"])]
    for c in c1["cases"]:
        norack = "disabled" in c["name"]
        ans = java_golden(a.cp, snapshot(topics, c["brokers"], {} if norack else c["racks"]), c["brokers"], norack=norack)
        regenerated["config1"].append({"name": c["name"], "answer": ans})
        if "fails" in c:
            f = ans.get("failed", {})
            check(c["name"], (f.get("topic"), f.get("message")),
                  (c1["topics"][c["fails"]["topic_index"]], "Partition %d could not be fully assigned!" % c["fails"]["partition"]),
                  "config1 (must fail)")
        else:
            for t, (name, _) in enumerate(topics):
                check(c["name"] + " / " + name, lists_of(ans, name), c["expected"][t], "config1")
    json.dump(regenerated, open(a.out, "w"), indent=1)
    print("wrote", os.path.relpath(a.out, ROOT), "-", len(bad), "vector(s) differ from tests/golden/survey_appendix_b.json")

    if a.bench > 0:
        d = tempfile.mkdtemp(prefix="kas_jvm_bench_")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "export_scenarios.py"), "--out", d, "--seed", "2026",
                               "--scenarios", str(a.bench), "--partitions", "100000", "--brokers", "1000", "--racks", "20",
                               "--rf", "3", "--actions", "remove1,remove_k,add_k,mixed"], stdout=subprocess.DEVNULL)
        files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".json"))
        r = subprocess.run(["java", "-Xms4g", "-Xmx4g", "-cp", a.cp, "siftscience.kafka.tools.JavaRefBench"] + files,
                           capture_output=True, text=True, check=True)
        rate = [float(ln.rsplit(":", 1)[1]) for ln in r.stdout.splitlines() if ln.startswith("scenarios/s")][-1]
        print(json.dumps({"cpu_baseline": {
            "value": rate, "unit": "scenarios/s", "cores": 1, "kind": "reference",
            "sample": f"the reference's own KafkaTopicAssigner.generateAssignment on a JVM ({a.bench} scenarios of 100k partitions x 1k "
                      f"brokers x 20 racks, RF 3, bench action mix, seed 2026; tools/JavaRefBench.java, one thread, solve only)"}}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
