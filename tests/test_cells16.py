"""ABI v5: kas_solve_host16 — the host call with 16-bit cells (include/kas_abi.h).  cur / out travel as uint16 node
indices; the claim the header makes is that everything the reference's algorithm derives from broker ids is their ORDER
(KafkaAssignmentStrategy.java:73-99 sorted nodes, :188-200 processing order by position, :263-278 ties by position), so
that solving the index form of a batch and mapping the cells back through the sorted node table gives the lists of the
int32 call.

CPU (no GPU needed): the oracle on the index form of a batch against the oracle on the batch itself — lists equal cell
for cell after the lookup, records equal except the digest (which covers the cells as emitted) — over the odd inputs of
the hypothesis strategy (ragged rows, duplicate brokers, brokers that left, partitions != keys(cur), failures), seeded
batches of every action, multi-topic scenarios with and without a Context.
GPU: kas_solve_host16 against the oracle on the index form, bit for bit including the digests; tables large enough to
be cut into scenario ranges (widen / solve / narrow of different ranges overlap), pools that start at odd cells (the
16-byte side of the conversion kernels is aligned by hand), the what-if form (selected rows only), and the refusals."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings

from kafka_assigner_amd import abi
from kafka_assigner_amd import generator as G
from kafka_assigner_amd.flatten import (FlatBatch, Scenario, Topic, cells16_to_ids, flatten, index_form, to_cells16)
from oracle_lib import oracle_solve
from parity_util import assert_same_outputs
from test_emu_parity import _batch, _multi_topic_scenarios
from test_oracle_vs_literal import scenarios


def _as_cells16(fb, out_idx):
    """an int32 out pool of node indices as the uint16 pool a 16-bit call returns"""
    return np.where(out_idx < 0, abi.KAS_CELL16_NONE, out_idx).astype(np.uint16)


def _check_index_form_on_the_oracle(fb, what):
    want = oracle_solve(fb)
    got = oracle_solve(index_form(fb))
    T, S = fb.n_topics, fb.n_scenarios
    for name in ("status", "fail_partition", "moved_replicas", "moved_partitions"):
        np.testing.assert_array_equal(got.topic_results[name][:T], want.topic_results[name][:T], err_msg=f"{what} topic {name}")
    for name in ("status", "fail_topic", "fail_partition", "moved_replicas", "moved_partitions"):
        np.testing.assert_array_equal(got.scenario_results[name][:S], want.scenario_results[name][:S], err_msg=f"{what} scenario {name}")
    ids = cells16_to_ids(fb, _as_cells16(fb, got.out[:max(fb.out_len, 1)]))
    np.testing.assert_array_equal(ids[:fb.out_len], want.out[:fb.out_len], err_msg=f"{what}: lists after the lookup")
    np.testing.assert_array_equal(got.ctx, want.ctx, err_msg=f"{what}: Context counters (per node position: no lookup)")
    return want, got


@settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(scenarios())
def test_index_form_gives_the_same_lists_small_odd_inputs(sc):
    brokers, racks, topics = sc
    for want_ctx in (True, False):
        fb = flatten([Scenario(brokers=brokers, racks=racks, want_context=want_ctx,
                               topics=[Topic(n, c, rf, parts) for n, c, rf, parts in topics])])
        _check_index_form_on_the_oracle(fb, "odd inputs")


@pytest.mark.parametrize("P,N,R,RF,actions", [
    (1000, 40, 8, 3, G.ACTIONS), (3000, 100, 10, 3, ("remove1",)), (2048, 64, 8, 2, ("add_k",)),
    (777, 40, 10, 5, G.ACTIONS), (640, 24, 8, 4, ("replace1", "remove1")),
])
def test_index_form_gives_the_same_lists_seeded_batches(P, N, R, RF, actions):
    fb = _batch(4242, 5, P, N, R, RF, actions)
    want, _ = _check_index_form_on_the_oracle(fb, "seeded batch")
    assert (want.scenario_results["moved_replicas"] > 0).any()
    # sparse, non-contiguous broker ids (id != index everywhere) and rack awareness off
    fb2 = _batch(4243, 3, P, N, R, RF, actions, rack_aware=False)
    fb2.node_id[:] = fb2.node_id * 7 + 1000
    fb2.cur[:] = np.where(fb2.cur >= 0, fb2.cur * 7 + 1000, fb2.cur)
    _check_index_form_on_the_oracle(fb2, "sparse ids")


def test_index_form_multi_topic_scenarios_and_round_trip_of_the_cells():
    fb = _multi_topic_scenarios(77, 3, 3, 700, 40, 8, 3)
    want, _ = _check_index_form_on_the_oracle(fb, "multi-topic")
    assert (want.topic_results["status"] != abi.KAS_OK).any()          # failing and skipped topics included
    c16 = to_cells16(fb)
    assert c16.dtype == np.uint16 and c16.shape == fb.cur.shape
    gone = c16 == abi.KAS_CELL16_NONE
    assert gone.any() and not gone.all()                               # brokers that left the set
    # every other cell names the broker it stood for
    owner = np.repeat(np.arange(fb.n_topics), fb.topics["n_partitions"].astype(np.int64) * fb.topics["cur_width"])
    scen_of_topic = np.repeat(np.arange(fb.n_scenarios), fb.scen["topic_count"])
    off = fb.scen["node_off"][scen_of_topic[owner]]
    np.testing.assert_array_equal(fb.node_id[(off + c16)[~gone]], fb.cur[~gone])
    # a cur table shared by scenarios with different broker sets (the what-if layout) has no node-index form
    from kafka_assigner_amd.flatten import node_set_batch
    cur = G.random_assignment(1, 500, 30, 5, 3)
    sets = [G.scenario_action(3, s, 30, 5, actions=("remove1",))[1] for s in range(3)]
    shared = node_set_batch([b.node_id for b in sets], [b.node_rack for b in sets], 500, 3, 3, shared_cur=True, cur=cur)
    with pytest.raises(ValueError):
        to_cells16(shared)


# ---------------------------------------------------------------------------------------------------------------------
def _check_hip16(fb, what, ctx=None, pinned=False):
    from kafka_assigner_amd import native
    from kafka_assigner_amd.flatten import host_tables16
    want = oracle_solve(index_form(fb), threads=0)
    keep = None
    if pinned:
        cur16 = native.PinnedArray(fb.cur.shape[0], np.uint16)
        cur16.array[:] = to_cells16(fb)
        t, ho = host_tables16(fb, cur16.array)
        out16 = native.PinnedArray(max(fb.out_len, 1), np.uint16)
        out16.array[:] = 0xFFFE
        ho.out = out16.array
        t.out = out16.array.ctypes.data
        keep = (cur16, out16)
        got = native.solve_host16(fb, ctx, tables=t, ho=ho)
    else:
        got = native.solve_host16(fb, ctx)
    assert got.out.dtype == np.uint16
    want.out = _as_cells16(fb, want.out)
    assert_same_outputs(fb, want, got, what)                           # (digests included: both cover node indices)
    ids = cells16_to_ids(fb, got.out)
    if keep:
        got.out = got.out.copy()
        for k in keep:
            k.close()
    return want, got, ids


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(scenarios())
def test_hip16_equals_oracle_small_odd_inputs(sc):
    brokers, racks, topics = sc
    fb = flatten([Scenario(brokers=brokers, racks=racks, want_context=True,
                           topics=[Topic(n, c, rf, parts) for n, c, rf, parts in topics])])
    _, _, ids = _check_hip16(fb, "hip 16-bit cells, odd inputs")
    np.testing.assert_array_equal(ids[:fb.out_len], oracle_solve(fb).out[:fb.out_len])


@pytest.mark.gpu
@pytest.mark.parametrize("P,N,R,RF,actions", [
    (1000, 40, 8, 3, G.ACTIONS), (3000, 100, 10, 3, ("remove1",)), (777, 40, 10, 5, G.ACTIONS),
    (8000, 80, 8, 3, ("replace1", "add_k")),
])
def test_hip16_equals_oracle_seeded_batches_and_the_int32_call(P, N, R, RF, actions):
    from kafka_assigner_amd import native
    fb = _batch(1234, 6, P, N, R, RF, actions)
    _, got, ids = _check_hip16(fb, "hip 16-bit cells")
    plain = native.solve_host(fb)                                      # the int32 call on the same batch: same lists
    np.testing.assert_array_equal(ids[:fb.out_len], plain.out[:fb.out_len])
    for name in ("status", "fail_partition", "moved_replicas", "moved_partitions"):
        np.testing.assert_array_equal(got.scenario_results[name], plain.scenario_results[name])
    # multi-topic scenarios with a Context handed in and back
    scs = []
    for s in range(3):
        cur = G.random_assignment(300 + s, 2500, 40, 8, 3)
        _, bs = G.scenario_action(300, s, 40, 8, actions=("remove1", "add_k"), max_add=4)
        racks = {int(b): "r%d" % int(r) for b, r in zip(bs.node_id, bs.node_rack)}
        scs.append(Scenario(brokers=[int(b) * 3 + 7 for b in bs.node_id], racks={b * 3 + 7: r for b, r in racks.items()}, want_context=True,
                            topics=[Topic("topic-%d" % t, {p: [int(x) * 3 + 7 for x in cur[p]] for p in range(2500 - 100 * t)}, 3) for t in range(3)]))
    fbm = flatten(scs)
    assert (fbm.topics["cur_off"] % 8 != 0).any()                     # pools that start at odd cells
    _, _, idsm = _check_hip16(fbm, "hip 16-bit cells, multi-topic with a Context")
    np.testing.assert_array_equal(idsm[:fbm.out_len], native.solve_host(fbm).out[:fbm.out_len])


@pytest.mark.gpu
def test_hip16_large_tables_are_cut_into_ranges_pinned_and_pageable():
    """120 scenarios x 100,000 x 3 cells: 72 MB up and 72 MB down at 2 bytes a cell -> scenario ranges whose upload, widen +
    solve + narrow and download overlap (kas_solve_host's pipeline); caller buffers pageable and from kas_host_alloc."""
    from kafka_assigner_amd import native
    from kafka_assigner_amd.flatten import node_set_batch
    S, P, N, R = 120, 100000, 1000, 10
    base = [G.random_assignment(50 + k, P, N, R, 3) for k in range(4)]
    cur = np.stack([base[s % 4] for s in range(S)])
    sets = [G.scenario_action(9, s, N, R, actions=G.BENCH_ACTIONS, max_add=50)[1] for s in range(S)]
    fb = node_set_batch([b.node_id for b in sets], [b.node_rack for b in sets], P, 3, 3, cur=cur)
    ctx = native.DeviceContext(0)
    want, got, ids = _check_hip16(fb, "hip 16-bit cells, 120 x 100k x 3", ctx)
    assert (want.scenario_results["status"] == abi.KAS_OK).sum() > 100
    _check_hip16(fb, "hip 16-bit cells, pinned caller buffers", ctx, pinned=True)
    plain = native.solve_host(fb, ctx)
    np.testing.assert_array_equal(ids[:fb.out_len], plain.out[:fb.out_len])
    ctx.close()


@pytest.mark.gpu
def test_hip16_select_returns_every_record_and_the_selected_rows_and_the_refusals():
    from kafka_assigner_amd import native
    fb = _batch(88, 7, 3000, 60, 6, 3, G.ACTIONS)
    want = oracle_solve(index_form(fb))
    sel = [5, 0, 3]
    got = native.solve_host16(fb, select=sel)
    for name in ("status", "fail_topic", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
        np.testing.assert_array_equal(got.scenario_results[name][:7], want.scenario_results[name][:7])
    at = 0
    for s in sel:
        td = fb.topics[int(fb.scen["topic_begin"][s])]
        lo, n = int(td["out_off"]), int(td["n_partitions"]) * int(td["out_width"])
        np.testing.assert_array_equal(got.out[at:at + n], _as_cells16(fb, want.out[lo:lo + n]))
        at += n
    assert at == native.selected_out_len(fb, sel) and got.out.shape[0] == at
    none = native.solve_host16(fb, select=[])
    np.testing.assert_array_equal(none.scenario_results["digest"][:7], want.scenario_results["digest"][:7])
    # refusals: a scenario index out of range, NULL tables
    with pytest.raises(native.KasError) as e:
        native.solve_host16(fb, select=[7])
    assert e.value.code == abi.KAS_E_INVALID_ARG
    from kafka_assigner_amd.flatten import host_tables16
    t, ho = host_tables16(fb, to_cells16(fb))
    t.cur = None
    with pytest.raises(native.KasError) as e:
        native.solve_host16(fb, tables=t, ho=ho)
    assert e.value.code == abi.KAS_E_INVALID_ARG
