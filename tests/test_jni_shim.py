"""The C side of the JNI shim (kafka-assigner_amd/host/jni/kas_jni.cpp), compiled against a stub
<jni.h> (tests/jni_stub: there is no JDK in this image) and driven from ctypes with buffers laid
out exactly as NativeAssignmentStrategy.java lays them out.  -m gpu: the shim calls kas_solve_host."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from kafka_assigner_amd import build as kbuild
from kafka_assigner_amd.flatten import Scenario, Topic, flatten, java_string_hashcode, unflatten_topic
from oracle_lib import oracle_solve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "kafka-assigner_amd", "host", "jni", "kas_jni.cpp")
WIDTH, HEADER = 8, 8


class StubBuffer(C.Structure):
    _fields_ = [("address", C.c_void_p), ("capacity", C.c_longlong)]


@pytest.fixture(scope="module")
def shim():
    kbuild.build()
    so = os.path.join(ROOT, "tests", "jni_stub", "libkas_jni_stub.so")
    deps = [SRC, os.path.join(ROOT, "tests", "jni_stub", "jni.h"), kbuild.LIB]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "tests", "jni_stub"),
                               "-I" + os.path.join(ROOT, "include"), "-o", so, SRC, "-L" + kbuild.CSRC, "-lkas_hip",
                               "-Wl,-rpath," + kbuild.CSRC, "-Wl,--allow-shlib-undefined"])
    import torch  # noqa: F401  (its HIP runtime must be the one in the process, see native.load)
    L = C.CDLL(so)
    fn = L.Java_siftscience_kafka_tools_NativeAssignmentStrategy_solveBatch
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(StubBuffer), C.POINTER(StubBuffer)]
    return fn


def _java_layout(topic, cur, racks, nodes, partitions, rf, counters):
    """What NativeAssignmentStrategy.solve() puts into the input ByteBuffer."""
    node_ids = sorted(nodes)
    rows = sorted(cur)
    n, p = len(node_ids), len(rows)
    cw = max([len(v) for v in cur.values()] + [0])
    ow = max(cw, rf, 1)
    rack_index, node_rack = {}, []
    for b in node_ids:
        r = racks.get(b, str(b))
        node_rack.append(rack_index.setdefault(r, len(rack_index)))
    buf = [java_string_hashcode(topic), p, cw, rf, ow, n, 1 if counters is not None else 0, 0]
    buf += node_ids + node_rack + rows + [len(cur[q]) for q in rows] + [1 if q in partitions else 0 for q in rows]
    for q in rows:
        buf += list(cur[q]) + [-1] * (cw - len(cur[q]))
    for b in node_ids:
        buf += [(counters or {}).get(b, {}).get(k, 0) for k in range(WIDTH)]
    return np.asarray(buf, dtype=np.int32), (n, p, cw, ow, node_ids, rows)


def test_shim_rejects_short_buffers_without_a_gpu(shim):
    inp = np.asarray([0, 4, 2, 2, 2, 3, 0, 0], dtype=np.int32)        # header only: tables missing
    out = np.zeros(4, dtype=np.int32)
    bi = StubBuffer(inp.ctypes.data, inp.nbytes)
    bo = StubBuffer(out.ctypes.data, out.nbytes)
    assert shim(None, None, C.byref(bi), C.byref(bo)) == -1                # KAS_E_INVALID_ARG


@pytest.mark.gpu
def test_shim_solves_the_junit_cluster_and_carries_the_context(shim):
    cur = {0: [10, 11], 1: [11, 12], 2: [12, 10], 3: [10, 12]}
    racks = {10: "a", 11: "b", 12: "c", 13: "a", 14: "b"}
    nodes = [10, 11, 12, 13, 14]
    counters = {}
    for topic in ("test", "topic-1"):                                      # one Context across two calls
        inp, (n, p, cw, ow, node_ids, rows) = _java_layout(topic, cur, racks, nodes, set(cur), 2, counters)
        out = np.zeros(4 + p * ow + n * WIDTH, dtype=np.int32)
        bi, bo = StubBuffer(inp.ctypes.data, inp.nbytes), StubBuffer(out.ctypes.data, out.nbytes)
        assert shim(None, None, C.byref(bi), C.byref(bo)) == 0
        fb = flatten([Scenario(brokers=nodes, racks=racks, context=counters, want_context=True,
                               topics=[Topic(topic, cur, 2)])])
        want = oracle_solve(fb)
        assert out[0] == want.topic_results["status"][0] == 0
        assert out[2] == want.topic_results["moved_replicas"][0]
        got = {rows[i]: [int(b) for b in out[4 + i * ow:4 + (i + 1) * ow] if b >= 0] for i in range(p)}
        assert got == unflatten_topic(fb, want.out, 0)
        ctx = out[4 + p * ow:].reshape(n, WIDTH)
        np.testing.assert_array_equal(ctx, want.ctx[:n * WIDTH].reshape(n, WIDTH))
        counters = {b: {k: int(ctx[i, k]) for k in range(WIDTH) if ctx[i, k]} for i, b in enumerate(node_ids)}
