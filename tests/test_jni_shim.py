"""The C side of the JNI shim (kafka-assigner_amd/host/jni/kas_jni.cpp), compiled against a stub
<jni.h> (tests/jni_stub: there is no JDK in this image) and driven from ctypes with buffers laid
out exactly as NativeAssignmentStrategy.java lays them out.  -m gpu: the shim calls kas_solve_host."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from kafka_assigner_amd import abi, build as kbuild
from kafka_assigner_amd.flatten import Scenario, Topic, flatten, java_string_hashcode, unflatten_topic
from oracle_lib import oracle_solve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "kafka-assigner_amd", "host", "jni", "kas_jni.cpp")
WIDTH, HEADER = 8, 12


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


def _payload(fb, select=None, device=0):
    """The batch payload NativeAssignmentStrategy.solveScenarios() writes (layout 3): header (with the
    device and the number of selected scenarios, -1 = every row comes back), scenario and topic
    descriptors as the C structs, node pools, cur, aux, ctx, select."""
    from kafka_assigner_amd.native import selected_out_len
    S, T = fb.n_scenarios, fb.n_topics
    cur = fb.cur if fb.cur.size else np.zeros(0, np.int32)
    sel = np.zeros(0, np.int32) if select is None else np.asarray(select, dtype=np.int32)
    ret_len = fb.out_len if select is None else selected_out_len(fb, sel)
    hdr = np.asarray([3, S, T, fb.node_id.size, cur.size, fb.aux.size, fb.ctx.size, ret_len,
                      device, -1 if select is None else sel.size, 0, 0], dtype=np.int32)
    parts = [hdr, fb.scen[:S].view(np.int32).reshape(-1), fb.topics[:T].view(np.int32).reshape(-1),
             fb.node_id, fb.node_rack, cur, fb.aux, fb.ctx, sel]
    inp = np.concatenate([np.ascontiguousarray(x, dtype=np.int32).reshape(-1) for x in parts])
    out = np.full(4 * T + 8 * S + ret_len + fb.ctx.size, -7, dtype=np.int32)
    return inp, out


def _payload16(fb, select=None, device=0):
    """Layout 4 with cellBits = 16 (ABI v5): cur[] / out[] as uint16 node indices, two to an int; everything else as
    layout 3.  Returns (in, out, ints of out[] that hold the rows)."""
    from kafka_assigner_amd.flatten import to_cells16
    from kafka_assigner_amd.native import selected_out_len
    S, T = fb.n_scenarios, fb.n_topics
    c16 = to_cells16(fb) if fb.cur.size else np.zeros(0, np.uint16)
    if c16.size % 2:
        c16 = np.concatenate([c16, np.zeros(1, np.uint16)])
    sel = np.zeros(0, np.int32) if select is None else np.asarray(select, dtype=np.int32)
    ret_len = fb.out_len if select is None else selected_out_len(fb, sel)
    hdr = np.asarray([4, S, T, fb.node_id.size, fb.cur.size if fb.cur.size else 0, fb.aux.size, fb.ctx.size, ret_len,
                      device, -1 if select is None else sel.size, 16, 0], dtype=np.int32)
    parts = [hdr, fb.scen[:S].view(np.int32).reshape(-1), fb.topics[:T].view(np.int32).reshape(-1),
             fb.node_id, fb.node_rack, c16.view(np.int32), fb.aux, fb.ctx, sel]
    inp = np.concatenate([np.ascontiguousarray(x, dtype=np.int32).reshape(-1) for x in parts])
    ret_ints = (ret_len + 1) // 2
    out = np.full(4 * T + 8 * S + ret_ints + fb.ctx.size, -7, dtype=np.int32)
    return inp, out, ret_ints


def _call(shim, inp, out):
    bi, bo = StubBuffer(inp.ctypes.data, inp.nbytes), StubBuffer(out.ctypes.data, out.nbytes)
    return shim(None, None, C.byref(bi), C.byref(bo))


def _unpack(fb, out, ret_len=None):
    S, T = fb.n_scenarios, fb.n_topics
    ret_len = fb.out_len if ret_len is None else ret_len
    tr = out[:4 * T].view(abi.TOPIC_RESULT_DTYPE)
    sr = out[4 * T:4 * T + 8 * S].view(abi.SCENARIO_RESULT_DTYPE)
    rows = out[4 * T + 8 * S:4 * T + 8 * S + ret_len]
    ctx = out[4 * T + 8 * S + ret_len:]
    return tr, sr, rows, ctx


def test_shim_rejects_short_buffers_and_unknown_layouts_without_a_gpu(shim):
    out = np.zeros(64, dtype=np.int32)
    hdr_only = np.asarray([3, 1, 1, 5, 8, 12, 0, 8, 0, -1, 0, 0], dtype=np.int32)   # tables missing
    assert _call(shim, hdr_only, out) == -1                                    # KAS_E_INVALID_ARG
    old_layout = np.zeros(64, dtype=np.int32); old_layout[0] = 2               # layout 2: 8-int header, no device
    assert _call(shim, old_layout, out) == -1
    neg = np.asarray([3, -1, 0, 0, 0, 0, 0, 0, 0, -1, 0, 0], dtype=np.int32)
    assert _call(shim, neg, out) == -1
    bad_dev = np.asarray([3, 0, 0, 0, 0, 0, 0, 0, -2, -1, 0, 0], dtype=np.int32)
    assert _call(shim, bad_dev, out) == -1
    bad_cells = np.asarray([4, 0, 0, 0, 0, 0, 0, 0, 0, -1, 8, 0], dtype=np.int32)    # layout 4: cellBits must be 0, 16 or 32
    assert _call(shim, bad_cells, out) == -1
    short16 = np.asarray([4, 1, 1, 5, 9, 0, 0, 9, 0, -1, 16, 0], dtype=np.int32)     # 16-bit cells: tables missing
    assert _call(shim, short16, out) == -1
    future = np.asarray([5, 0, 0, 0, 0, 0, 0, 0, 0, -1, 0, 0], dtype=np.int32)
    assert _call(shim, future, out) == -1


@pytest.mark.gpu
def test_shim_solves_the_junit_cluster_and_carries_the_context(shim):
    """One scenario, one topic per call (what KTA:70-71 does), the Context carried across two calls."""
    cur = {0: [10, 11], 1: [11, 12], 2: [12, 10], 3: [10, 12]}
    racks = {10: "a", 11: "b", 12: "c", 13: "a", 14: "b"}
    nodes = [10, 11, 12, 13, 14]
    counters = {}
    for topic in ("test", "topic-1"):
        fb = flatten([Scenario(brokers=nodes, racks=racks, context=counters, want_context=True,
                               topics=[Topic(topic, cur, 2)])])
        inp, out = _payload(fb)
        assert _call(shim, inp, out) == 0
        want = oracle_solve(fb)
        tr, sr, rows, ctx = _unpack(fb, out)
        assert tr["status"][0] == want.topic_results["status"][0] == 0
        assert tr["moved_replicas"][0] == want.topic_results["moved_replicas"][0]
        np.testing.assert_array_equal(rows, want.out[:fb.out_len])
        np.testing.assert_array_equal(ctx, want.ctx[:fb.ctx.size])
        n = len(nodes)
        c2 = ctx.reshape(n, WIDTH)
        counters = {b: {k: int(c2[i, k]) for k in range(WIDTH) if c2[i, k]} for i, b in enumerate(sorted(nodes))}


@pytest.mark.gpu
def test_shim_batch_payload_many_scenarios_and_topics(shim):
    """The batch path through JNI: several scenarios, several topics each (with and without a
    Context, one scenario that strands and skips its later topics), one solveBatch call; then the
    same payload again — the second call must be served by the context's cached plan and buffers."""
    from kafka_assigner_amd import generator as G
    scs = []
    for s in range(5):
        act, bs = G.scenario_action(11, s, 60, 12, actions=("add_k", "remove1", "replace1"), max_add=6)
        racks = {int(b): "r%d" % int(r) for b, r in zip(bs.node_id, bs.node_rack)}
        topics = []
        for t in range(3):
            cur = G.random_assignment(3 + 7 * s + t, 700 + 13 * t, 60, 12, 3)
            topics.append(Topic("topic-%d" % t, {p: cur[p].tolist() for p in range(cur.shape[0])}, 3))
        scs.append(Scenario(brokers=[int(b) for b in bs.node_id], racks=racks, topics=topics,
                            want_context=(s % 2 == 0)))
    fb = flatten(scs)
    want = oracle_solve(fb)
    assert (want.topic_results["status"] == abi.KAS_OK).sum() >= 6
    for _ in range(2):
        inp, out = _payload(fb)
        assert _call(shim, inp, out) == 0
        tr, sr, rows, ctx = _unpack(fb, out)
        for f in ("status", "fail_partition", "moved_replicas", "moved_partitions"):
            np.testing.assert_array_equal(tr[f], want.topic_results[f][:fb.n_topics])
        for f in ("status", "fail_topic", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
            np.testing.assert_array_equal(sr[f], want.scenario_results[f][:fb.n_scenarios])
        np.testing.assert_array_equal(rows, want.out[:fb.out_len])
        ok_ctx = np.ones(fb.ctx.size, dtype=bool)
        np.testing.assert_array_equal(ctx[ok_ctx], want.ctx[:fb.ctx.size][ok_ctx])


@pytest.mark.gpu
def test_shim_what_if_payload_returns_records_for_all_and_rows_for_the_selected(shim):
    """Layout 3's what-if form: 24 broker-set variants over ONE shared current assignment (every topic
    descriptor points at the same cur rows), rows requested for two of them.  Records of all 24 and the
    rows of the two must be the oracle's; an out-of-range device index is refused."""
    from kafka_assigner_amd import generator as G
    from kafka_assigner_amd.native import selected_out_len
    from test_emu_parity import _batch
    P, N, R, RF, S = 20000, 200, 10, 3, 24
    fb = _batch(31, S, P, N, R, RF, G.BENCH_ACTIONS)
    fb.cur = G.random_assignment(31, P, N, R, RF).reshape(-1).copy()
    fb.topics["cur_off"] = 0
    want = oracle_solve(fb, threads=0)
    select = [3, 17]
    inp, out = _payload(fb, select=select)
    assert _call(shim, inp, out) == 0
    tr, sr, rows, _ = _unpack(fb, out, selected_out_len(fb, select))
    for f in ("status", "fail_partition", "moved_replicas", "moved_partitions"):
        np.testing.assert_array_equal(tr[f], want.topic_results[f][:S])
    for f in ("status", "fail_topic", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
        np.testing.assert_array_equal(sr[f], want.scenario_results[f][:S])
    cells = P * RF
    for k, s_ in enumerate(select):
        np.testing.assert_array_equal(rows[k * cells:(k + 1) * cells], want.out[s_ * cells:(s_ + 1) * cells])
    inp, out = _payload(fb, select=select, device=63)
    assert _call(shim, inp, out) == -1


@pytest.mark.gpu
def test_shim_layout_4_with_16_bit_cells_batch_and_selected_rows(shim):
    """Layout 4, cellBits = 16 (kas_solve_host16 behind the shim): the batch payload of several scenarios and topics with
    Contexts, and the form that returns only selected scenarios' rows — cells are node indices (0xFFFF pad), records and
    Context counters as in layout 3; mapped through the node tables the rows are the oracle's lists of the int32 batch.
    cellBits = 32 in a layout-4 header is the int32 payload."""
    from kafka_assigner_amd import generator as G
    from kafka_assigner_amd.flatten import cells16_to_ids, index_form
    from kafka_assigner_amd.native import selected_out_len
    scs = []
    for s in range(4):
        act, bs = G.scenario_action(21, s, 60, 12, actions=("add_k", "remove1", "replace1"), max_add=6)
        racks = {int(b) * 5 + 3: "r%d" % int(r) for b, r in zip(bs.node_id, bs.node_rack)}
        topics = []
        for t in range(3):
            cur = G.random_assignment(5 + 7 * s + t, 701 + 12 * t, 60, 12, 3)
            topics.append(Topic("topic-%d" % t, {p: [int(x) * 5 + 3 for x in cur[p]] for p in range(cur.shape[0])}, 3))
        scs.append(Scenario(brokers=[int(b) * 5 + 3 for b in bs.node_id], racks=racks, topics=topics, want_context=(s % 2 == 0)))
    fb = flatten(scs)
    want_ids = oracle_solve(fb)
    want = oracle_solve(index_form(fb))
    inp, out, ret_ints = _payload16(fb)
    assert _call(shim, inp, out) == 0
    S, T = fb.n_scenarios, fb.n_topics
    tr = out[:4 * T].view(abi.TOPIC_RESULT_DTYPE)
    sr = out[4 * T:4 * T + 8 * S].view(abi.SCENARIO_RESULT_DTYPE)
    rows16 = out[4 * T + 8 * S:4 * T + 8 * S + ret_ints].view(np.uint16)[:fb.out_len]
    ctx = out[4 * T + 8 * S + ret_ints:]
    for f in ("status", "fail_partition", "moved_replicas", "moved_partitions"):
        np.testing.assert_array_equal(tr[f], want.topic_results[f][:T])
    for f in ("status", "fail_topic", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
        np.testing.assert_array_equal(sr[f], want.scenario_results[f][:S])
    np.testing.assert_array_equal(rows16, np.where(want.out[:fb.out_len] < 0, 0xFFFF, want.out[:fb.out_len]).astype(np.uint16))
    np.testing.assert_array_equal(cells16_to_ids(fb, rows16), want_ids.out[:fb.out_len])
    np.testing.assert_array_equal(ctx, want_ids.ctx[:fb.ctx.size])
    # selected rows only
    select = [2, 1]
    inp, out, ret_ints = _payload16(fb, select=select)
    assert _call(shim, inp, out) == 0
    n_sel = selected_out_len(fb, select)
    sr = out[4 * T:4 * T + 8 * S].view(abi.SCENARIO_RESULT_DTYPE)
    np.testing.assert_array_equal(sr["digest"], want.scenario_results["digest"][:S])
    rows16 = out[4 * T + 8 * S:4 * T + 8 * S + ret_ints].view(np.uint16)[:n_sel]
    at = 0
    for s_ in select:
        for t in range(int(fb.scen["topic_begin"][s_]), int(fb.scen["topic_begin"][s_]) + int(fb.scen["topic_count"][s_])):
            td = fb.topics[t]
            lo, n = int(td["out_off"]), int(td["n_partitions"]) * int(td["out_width"])
            np.testing.assert_array_equal(rows16[at:at + n], np.where(want.out[lo:lo + n] < 0, 0xFFFF, want.out[lo:lo + n]).astype(np.uint16))
            at += n
    assert at == n_sel
    # a layout-4 header with int32 cells
    inp, out = _payload(fb)
    inp[0], inp[10] = 4, 32
    assert _call(shim, inp, out) == 0
    np.testing.assert_array_equal(_unpack(fb, out)[2], want_ids.out[:fb.out_len])
