"""tools/jvm_check.py (the recipe for a machine with a JDK) cannot run its Java half here — no JDK in the image.
What CAN be held to something: its plumbing.  With the JVM replaced by the literal Python restatement of the reference
(oracle/literal_ref.py, test infrastructure) every vector of tests/golden/survey_appendix_b.json must come back identical through
the script's own snapshot writer, argument builder and comparer — so that a difference reported on a real JVM is the JVM's."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("jvm_check", os.path.join(ROOT, "tools", "jvm_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _stand_in(cp, snap, brokers, desired_rf=-1, norack=False):
    """what tools/JavaGolden.java does, with oracle/literal_ref.py in the JVM's place"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import literal_ref as R
    racks = {} if norack else {int(b["id"]): b["rack"] for b in snap["brokers"] if "rack" in b}
    cur = {}
    for p in snap["partitions"]:
        cur.setdefault(p["topic"], {})[int(p["partition"])] = list(p["replicas"])
    assigner = R.KafkaTopicAssigner()
    out = {"version": 1, "partitions": []}
    for topic, m in cur.items():
        try:
            res = assigner.generate_assignment(topic, m, set(int(b) for b in brokers), {k: v for k, v in racks.items() if k in set(brokers)}, desired_rf)
        except (R.IllegalStateException, R.ArrayIndexOutOfBoundsException) as e:
            out["failed"] = {"topic": topic, "exception": type(e).__name__, "message": str(e)}
            break
        for p in sorted(res):
            out["partitions"].append({"topic": topic, "partition": p, "replicas": list(res[p])})
    return out


def test_jvm_check_plumbing_returns_every_golden_vector(tmp_path, monkeypatch, capsys):
    m = _load()
    monkeypatch.setattr(m, "java_golden", _stand_in)
    monkeypatch.setattr(sys, "argv", ["jvm_check.py", "--cp", "unused", "--bench", "0", "--out", str(tmp_path / "b.jvm.json")])
    rc = m.main()
    out = capsys.readouterr().out
    assert rc == 0, out
    assert "DIFFERENT" not in out and out.count("ok ") >= 4 + 4 + 13
    regenerated = json.load(open(tmp_path / "b.jvm.json"))
    assert len(regenerated["ktat"]) == 4 and len(regenerated["config1"]) == 5


def test_makefile_names_the_one_command():
    mk = open(os.path.join(ROOT, "tools", "Makefile")).read()
    assert "jvm-check:" in mk and "JavaGolden.java JavaRefBench.java" in mk and "jvm_check.py" in mk
