"""C++ host mirror + CLI (kafka-assigner_amd/host): `--mode PRINT_REASSIGNMENT` over a cluster
snapshot, the plumbing of BASELINE.json configs[0] (3 topics x 12 partitions, 6 brokers / 3
racks, RF 3).  The snapshot-only modes and argument handling run anywhere; the solve itself needs
the MI355X (-m gpu) and is compared with the known-answer vectors of SURVEY.md Appendix B as
PARSED JSON (the reference's key order is JVM-dependent, quirk Q11)."""
import json
import os
import subprocess

import pytest

from kafka_assigner_amd import build as kbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_appendix_b.json")))
C1 = G["config1"]


@pytest.fixture(scope="module")
def cli():
    return kbuild.build_host()


def _snapshot(tmp_path, brokers, racks, extra_brokers=()):
    snap = {"brokers": [], "partitions": []}
    for b in sorted(set(brokers) | set(extra_brokers)):
        e = {"id": b, "host": f"kafka-{b}.example.com", "port": 9092}
        if str(b) in racks or b in racks:
            e["rack"] = racks.get(str(b), racks.get(b))
        snap["brokers"].append(e)
    for t, topic in enumerate(C1["topics"]):
        for p, reps in sorted(C1["current"][t].items(), key=lambda kv: int(kv[0])):
            snap["partitions"].append({"topic": topic, "partition": int(p), "replicas": reps})
    path = tmp_path / "snapshot.json"
    path.write_text(json.dumps(snap))
    return str(path), snap


def _run(cli, *args, cells32=False):
    """cells32: KAS_CELLS32=1 — the mirror hands kas_solve_host int32 broker ids instead of kas_solve_host16 node indices"""
    env = dict(os.environ, KAS_CELLS32="1") if cells32 else None
    return subprocess.run([cli, *args], capture_output=True, text=True, timeout=120, env=env)


def _sections(stdout):
    """{'CURRENT ASSIGNMENT': json, 'NEW ASSIGNMENT': json, ...}"""
    out, lines = {}, stdout.splitlines()
    for i, line in enumerate(lines):
        if line.endswith(":") and i + 1 < len(lines):
            out[line[:-1]] = json.loads(lines[i + 1])
    return out


def test_usage_on_bad_arguments_returns_normally(cli, tmp_path):
    # KAG:263-270: arg problems print the usage and RETURN (exit code 0), quirk Q12
    r = _run(cli, "--mode", "PRINT_REASSIGNMENT")
    assert r.returncode == 0 and "kafka-assignment-generator.sh [options...]" in r.stderr
    path, _ = _snapshot(tmp_path, range(6), {})
    r = _run(cli, "--snapshot", path, "--mode", "PRINT_REASSIGNMENT", "--integer_broker_ids", "1",
             "--broker_hosts", "x")
    assert r.returncode == 0 and r.stdout == "" and "options" in r.stderr


def test_print_current_brokers_and_assignment_round_trip_the_snapshot(cli, tmp_path):
    case = C1["cases"][0]
    path, snap = _snapshot(tmp_path, case["brokers"], case["racks"])
    r = _run(cli, "--snapshot", path, "--mode", "PRINT_CURRENT_BROKERS")
    assert r.returncode == 0
    assert _sections(r.stdout)["CURRENT BROKERS"] == snap["brokers"]         # KAG:113-129 shape
    r = _run(cli, "--snapshot", path, "--mode", "PRINT_CURRENT_ASSIGNMENT", "--topics", "topic-1")
    cur = _sections(r.stdout)["CURRENT ASSIGNMENT"]
    assert cur["version"] == 1
    assert cur["partitions"] == [p for p in snap["partitions"] if p["topic"] == "topic-1"]


def test_unknown_hostnames_are_an_error_only_for_the_include_list(cli, tmp_path):
    path, _ = _snapshot(tmp_path, range(6), {})
    r = _run(cli, "--snapshot", path, "--mode", "PRINT_REASSIGNMENT", "--broker_hosts", "nope.example.com")
    assert r.returncode == 1 and "Some hostnames could not be found! We found: []" in r.stderr   # KAG:199-201


def _expected_new(case):
    parts = []
    for t, topic in enumerate(C1["topics"]):
        for p, reps in sorted(case["expected"][t].items(), key=lambda kv: int(kv[0])):
            parts.append({"topic": topic, "partition": int(p), "replicas": reps})
    return {"version": 1, "partitions": parts}


@pytest.mark.gpu
@pytest.mark.parametrize("cells32", [False, True], ids=["cells16", "cells32"])
@pytest.mark.parametrize("case", [c for c in C1["cases"] if "fails" not in c], ids=lambda c: c["name"])
def test_print_reassignment_matches_appendix_b(cli, tmp_path, case, cells32):
    all_brokers = set(range(9))
    racks = {str(b): "abc"[b % 3] for b in range(6)}
    racks.update({"6": "c", "7": "a", "8": "b"})             # extra brokers
    path, snap = _snapshot(tmp_path, all_brokers, racks)
    args = ["--snapshot", path, "--mode", "PRINT_REASSIGNMENT",
            "--integer_broker_ids", ",".join(str(b) for b in case["brokers"])]
    if not case["racks"]:
        args.append("--disable_rack_awareness")
    r = _run(cli, *args, cells32=cells32)
    assert r.returncode == 0, r.stderr
    sec = _sections(r.stdout)
    assert sec["CURRENT ASSIGNMENT"]["partitions"] == snap["partitions"]     # rollback aid, KAG:159-160
    assert sec["NEW ASSIGNMENT"] == _expected_new(case)


@pytest.mark.gpu
def test_print_reassignment_decommission_by_hostname_aborts_like_the_reference(cli, tmp_path):
    case = next(c for c in C1["cases"] if "fails" in c)
    live = set(case["brokers"]) | {5}
    path, _ = _snapshot(tmp_path, live, {str(b): "abc"[b % 3] for b in live})
    r = _run(cli, "--snapshot", path, "--mode", "PRINT_REASSIGNMENT",
             "--broker_hosts_to_remove", "kafka-5.example.com")
    # uncaught IllegalStateException after the rollback block, before any NEW ASSIGNMENT
    assert r.returncode == 1
    assert "CURRENT ASSIGNMENT:" in r.stdout and "NEW ASSIGNMENT" not in r.stdout
    assert "Partition %d could not be fully assigned!" % case["fails"]["partition"] in r.stderr


def test_without_a_gpu_the_solve_fails_loudly(cli, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    path, _ = _snapshot(tmp_path, range(6), {})
    r = _run(cli, "--snapshot", path, "--mode", "PRINT_REASSIGNMENT")
    assert r.returncode == 2 and "no HIP device" in r.stderr                  # no CPU fallback


def test_export_scenarios_writes_cli_snapshots(tmp_path):
    """tools/export_scenarios.py (input side of the JVM harness) writes snapshots the
    CLI reads back unchanged."""
    out = tmp_path / "scen"
    r = subprocess.run(["python", os.path.join(ROOT, "tools", "export_scenarios.py"), "--out", str(out),
                        "--scenarios", "2", "--partitions", "300", "--brokers", "20", "--racks", "5"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    snap = json.load(open(out / "scen_0000.json"))
    assert len(snap["partitions"]) == 300 and set(snap["solve_brokers"]) <= {b["id"] for b in snap["brokers"]}
    cli_path = kbuild.build_host()
    r = _run(cli_path, "--snapshot", str(out / "scen_0000.json"), "--mode", "PRINT_CURRENT_ASSIGNMENT")
    assert r.returncode == 0
    assert _sections(r.stdout)["CURRENT ASSIGNMENT"]["partitions"] == snap["partitions"]


@pytest.mark.gpu
def test_exported_scenarios_through_the_cli_equal_the_oracle(cli, tmp_path):
    """The path a JVM site would diff against JavaGolden: exported snapshot -> CLI -> NEW ASSIGNMENT,
    here against the oracle on the same inputs (list-equal, or the same failing partition)."""
    import numpy as np
    from kafka_assigner_amd.flatten import Scenario, Topic, flatten
    from oracle_lib import oracle_solve
    from kafka_assigner_amd import abi
    out = tmp_path / "scen"
    r = subprocess.run(["python", os.path.join(ROOT, "tools", "export_scenarios.py"), "--out", str(out),
                        "--scenarios", "4", "--partitions", "2000", "--brokers", "40", "--racks", "8"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    for s in range(4):
        path = str(out / f"scen_{s:04d}.json")
        snap = json.load(open(path))
        cur = {p["partition"]: p["replicas"] for p in snap["partitions"]}
        racks = {b["id"]: b["rack"] for b in snap["brokers"]}
        fb = flatten([Scenario(brokers=snap["solve_brokers"], racks=racks, want_context=False,
                               topics=[Topic("t0", cur, 3, None)])])
        want = oracle_solve(fb)
        r = _run(cli, "--snapshot", path, "--mode", "PRINT_REASSIGNMENT",
                 "--integer_broker_ids", ",".join(str(b) for b in snap["solve_brokers"]), cells32=(s == 3))
        if want.scenario_results["status"][0] == abi.KAS_OK:
            assert r.returncode == 0, r.stderr
            new = _sections(r.stdout)["NEW ASSIGNMENT"]["partitions"]
            got = np.array([p["replicas"] for p in new], dtype=np.int32)
            np.testing.assert_array_equal(got.reshape(-1), want.out[:fb.out_len])
        else:
            assert r.returncode == 1
            assert "Partition %d could not be fully assigned!" % want.scenario_results["fail_partition"][0] in r.stderr
