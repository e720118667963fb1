#!/usr/bin/env python
"""tests/golden/make_golden.py — where the golden vectors come from, and a way to re-derive them.

tests/golden/survey_appendix_b.json was transcribed BY HAND from SURVEY.md Appendix B, whose
vectors the survey derived with its own scratch restatement of the reference (one of them traced
line by line against KafkaAssignmentStrategy.java; all of them satisfy every assertion of the
reference's KafkaTopicAssignerTest.java:18-187).  Nothing in this repository generated that file.

This script is the committed generator the fixture never had: it recomputes every `expected` /
`moved_replicas` / `final_context` entry of the JSON with the line-by-line Python restatement
(oracle/literal_ref.py — the closest thing to running the Java that this image allows: no JDK)
and either reports the differences (default: none, exit 0) or, with --write, rewrites the
file from the recomputation.  tools/JavaGolden.java + tools/compare_golden.py do the same against
the real reference classes wherever a JDK exists.

usage: python tests/golden/make_golden.py [--write]
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import literal_ref  # noqa: E402

PATH = os.path.join(HERE, "survey_appendix_b.json")


def ik(d):
    return {int(k): v for k, v in d.items()}


def sk(d):
    return {str(k): v for k, v in sorted(d.items())}


def solve(topic, cur, brokers, racks, rf, assigner=None):
    a = assigner or literal_ref.KafkaTopicAssigner()
    return a.generate_assignment(topic, ik(cur), set(brokers), ik(racks), rf), a


def main():
    g = json.load(open(PATH))
    new = json.loads(json.dumps(g))
    diffs = []
    for i, c in enumerate(g["ktat"]):
        out, _ = solve(c["topic"], c["current"], c["brokers"], c["racks"], c["desired_rf"])
        new["ktat"][i]["expected"] = sk(out)
        if sk(out) != c["expected"]:
            diffs.append(f"ktat/{c['name']}")
    c1 = g["config1"]
    for i, c in enumerate(c1["cases"]):
        if "fails" in c:
            try:
                solve(c1["topics"][0], c1["current"][0], c["brokers"], c["racks"], -1)
                diffs.append(f"config1/{c['name']}: expected a failure")
            except literal_ref.IllegalStateException as e:
                if str(e) != "Partition %d could not be fully assigned!" % c["fails"]["partition"]:
                    diffs.append(f"config1/{c['name']}: {e}")
            continue
        a = literal_ref.KafkaTopicAssigner()
        exp, moved = [], []
        for t, topic in enumerate(c1["topics"]):
            out, a = solve(topic, c1["current"][t], c["brokers"], c["racks"], -1, a)
            exp.append(sk(out))
            cur = ik(c1["current"][t])
            moved.append(sum(len(set(out[p]) - set(cur[p])) for p in cur))
        new["config1"]["cases"][i]["expected"] = exp
        new["config1"]["cases"][i]["moved_replicas"] = moved
        if exp != c["expected"] or moved != c["moved_replicas"]:
            diffs.append(f"config1/{c['name']}")
        if "final_context" in c:
            ctx = {str(n): {str(r): v for r, v in sorted(m.items()) if v} for n, m in
                   sorted(a.assignment_context.counter.items()) if any(m.values())}
            new["config1"]["cases"][i]["final_context"] = ctx
            if ctx != c["final_context"]:
                diffs.append(f"config1/{c['name']}: final_context")
    for i, c in enumerate(g["quirks"]):
        try:
            out, _ = solve(c["topic"], c["current"], c["brokers"], c["racks"], c["desired_rf"])
            if c.get("error") == "index":
                diffs.append(f"quirks/{c['name']}: expected the KAS:190 index error")
            else:
                new["quirks"][i]["expected"] = sk(out)
                if sk(out) != c["expected"]:
                    diffs.append(f"quirks/{c['name']}")
        except literal_ref.ArrayIndexOutOfBoundsException:
            if c.get("error") != "index":
                diffs.append(f"quirks/{c['name']}: unexpected index error")
    if "--write" in sys.argv:
        json.dump(new, open(PATH, "w"), indent=1)
        print("rewrote", PATH, "-", len(diffs), "entries changed")
        return 0
    print("golden vectors re-derived with oracle/literal_ref.py:", "all identical" if not diffs else diffs)
    return 1 if diffs else 0


if __name__ == "__main__":
    sys.exit(main())
