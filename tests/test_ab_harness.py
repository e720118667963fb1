"""tools/ab_harness.cpp (A/B of library builds on seeded batches): the batches it generates are what it says they
are.  Compiles the tool (hipcc; host code only is exercised here), lets it dump a batch and run it through the emulator of the
kernel source (AB_EMU), and solves the dumped tables with the oracle: status, movement and digest of every scenario must agree —
so a checksum the tool prints on the GPU stands for oracle-checked results."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from kafka_assigner_amd import abi
from kafka_assigner_amd.flatten import FlatBatch
from emu_lib import build_emu
from oracle_lib import oracle_solve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
HASHES = {1: (3644,), 3: (-1139260654, -1139260653, -1139260652)}


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    exe = str(tmp_path_factory.mktemp("abh") / "ab_harness")
    subprocess.check_call([HIPCC, "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tools", "ab_harness.cpp"), "-ldl"], stderr=subprocess.DEVNULL)
    return exe


@pytest.mark.parametrize("mode,S", [("shape:3000:60:6:3", 3), ("c3mix", 1), ("multi:1500:40:8", 2), ("shape:1200:50:10:5", 2)])
def test_harness_batches_through_the_emulator_equal_the_oracle(harness, tmp_path, mode, S):
    if mode == "c3mix":
        S = 4                                                # one scenario of every action kind (100k x 1k each: ~2 s)
    dump = str(tmp_path / "batch.bin")
    r = subprocess.run([harness, mode, str(S), "1", "x"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, AB_EMU=build_emu(), AB_DUMP=dump))
    assert r.returncode == 0, r.stdout + r.stderr
    got = {int(m.group(1)): (int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5), 16)) for m in re.finditer(
        r"scenario (\d+): status (-?\d+) moved_replicas (-?\d+) moved_partitions (-?\d+) digest ([0-9a-f]+)", r.stdout)}
    assert len(got) == S
    raw = np.fromfile(dump, dtype=np.int32)
    S_, T, P = (int(x) for x in raw[:3])
    rfs = [int(x) for x in raw[3:3 + T]]
    n_nodes = raw[6:6 + S_].astype(np.int64)
    pool = int(n_nodes.sum())
    node_id, node_rack = raw[6 + S_:6 + S_ + pool], raw[6 + S_ + pool:6 + S_ + 2 * pool]
    cur = raw[6 + S_ + 2 * pool:]
    assert S_ == S and cur.shape[0] == S * P * sum(rfs)
    scen = np.zeros(S, dtype=abi.SCENARIO_DESC_DTYPE)
    topics = np.zeros(S * T, dtype=abi.TOPIC_DESC_DTYPE)
    off, toff = 0, 0
    for s in range(S):
        scen[s] = (int(n_nodes[s]), s * T, T, 0, off, -1)
        off += int(n_nodes[s])
        for k in range(T):
            topics[s * T + k] = (HASHES[T][k], P, rfs[k], rfs[k], rfs[k], 0, toff, toff, -1, -1, -1)
            toff += P * rfs[k]
    fb = FlatBatch(scen=scen, topics=topics, node_id=node_id.copy(), node_rack=node_rack.copy(), cur=cur.copy(),
                   aux=np.zeros(0, np.int32), ctx=np.zeros(0, np.int32), out_len=int(cur.shape[0]))
    want = oracle_solve(fb).scenario_results
    for s in range(S):
        assert got[s] == (int(want["status"][s]), int(want["moved_replicas"][s]), int(want["moved_partitions"][s]),
                          int(want["digest"][s])), f"{mode} scenario {s}"
