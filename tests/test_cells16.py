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


# ---- the kernels' own 16-bit I/O (kas_plan_create16 / kas_solve_device16) on the CPU emulator -------------------------
def _want16(fb):
    want = oracle_solve(index_form(fb))
    want.out = _as_cells16(fb, want.out)
    return want


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(scenarios())
def test_emu16_equals_oracle_small_odd_inputs(sc):
    """Ragged rows, duplicate brokers, brokers that left, partitions != keys(cur), empty topics, failures — with and without
    a Context; batches with lists more than 3 wide are refused (KAS_E_UNSUPPORTED: the caller widens)."""
    from emu_lib import emu_solve16
    brokers, racks, topics = sc
    for want_ctx in (True, False):
        fb = flatten([Scenario(brokers=brokers, racks=racks, want_context=want_ctx,
                               topics=[Topic(n, c, rf, parts) for n, c, rf, parts in topics])])
        wide = bool((fb.topics["out_width"] > 3).any())
        try:
            got = emu_solve16(fb)
        except RuntimeError as e:
            assert "rc=-3" in str(e) and wide, str(e)
            continue
        assert not wide
        assert_same_outputs(fb, _want16(fb), got, "emu, 16-bit cells, odd inputs")
        assert_same_outputs(fb, _want16(fb), emu_solve16(fb, flags=2), "emu, 16-bit cells, round form")


@pytest.mark.parametrize("P,N,R,RF,actions", [
    (1000, 40, 8, 3, G.ACTIONS), (3000, 100, 10, 3, ("remove1",)), (2048, 64, 8, 2, ("add_k",)),
    (8000, 80, 8, 3, ("replace1", "add_k")),
])
def test_emu16_equals_oracle_seeded_batches_every_plan_variant(P, N, R, RF, actions):
    from emu_lib import (FILL_WITH_P4, NO_RTN_QUOTA, RELAX_TILES_64, RELAX_TILES_128, TICKET_ORDER, emu_solve16,
                         last_split_p4)
    fb = _batch(1234, 6, P, N, R, RF, actions)
    want = _want16(fb)
    assert_same_outputs(fb, want, emu_solve16(fb), "emu 16-bit cells")
    assert last_split_p4() == 1
    for flags, what in ((FILL_WITH_P4, "first fit inside the fill workgroup"), (RELAX_TILES_64, "tiles of 64 rows"),
                        (RELAX_TILES_128, "double tiles"), (RELAX_TILES_64 | NO_RTN_QUOTA, "quota without the atomic-with-return"),
                        (1, "general fill"), (2, "round form"), (TICKET_ORDER, "ticket form asked for: round form"),
                        (RELAX_TILES_64 | (1 << 8), "one fill wavefront"), (RELAX_TILES_64 | (2 << 8), "two fill wavefronts"),
                        (8, "chunk-count pass"), (3 | (1 << 8), "general fill + round form, one wavefront")):
        assert_same_outputs(fb, want, emu_solve16(fb, flags=flags), "emu 16-bit cells, " + what)
    # sparse, non-contiguous broker ids: the cells do not care
    fb2 = _batch(77, 3, P, N, R, RF, actions, rack_aware=False)
    fb2.node_id[:] = fb2.node_id * 7 + 1000
    fb2.cur[:] = np.where(fb2.cur >= 0, fb2.cur * 7 + 1000, fb2.cur)
    assert_same_outputs(fb2, _want16(fb2), emu_solve16(fb2), "emu 16-bit cells, sparse ids, rack awareness off")
    # rows that are not rack-diverse: the general fill inside the fill kernel
    fb3 = _batch(78, 3, P, N, 2, RF, ("remove1", "add_k"), cyclic=True) if RF <= 2 else _batch(78, 3, P, N, R, RF, ("remove1", "add_k"), cyclic=True)
    assert_same_outputs(fb3, _want16(fb3), emu_solve16(fb3), "emu 16-bit cells, cyclic rows")


def test_emu16_multi_topic_scenarios_with_and_without_a_context_and_refusals():
    from emu_lib import RELAX_TILES_64, VERIFY_SAMPLE, emu_solve16
    fb = _multi_topic_scenarios(77, 3, 3, 700, 40, 8, 3)
    want = _want16(fb)
    assert (want.topic_results["status"] != abi.KAS_OK).any()          # failing and skipped topics: rows of padding
    for flags in (0, RELAX_TILES_64, 2):
        assert_same_outputs(fb, want, emu_solve16(fb, flags=flags), "emu 16-bit cells, multi-topic, flags %#x" % flags)
    scs = []
    for s in range(3):
        cur = G.random_assignment(300 + s, 2500, 40, 8, 3)
        _, bs = G.scenario_action(300, s, 40, 8, actions=("remove1", "add_k"), max_add=4)
        racks = {int(b) * 3 + 7: "r%d" % int(r) for b, r in zip(bs.node_id, bs.node_rack)}
        scs.append(Scenario(brokers=[int(b) * 3 + 7 for b in bs.node_id], racks=racks, want_context=True,
                            topics=[Topic("topic-%d" % t, {p: [int(x) * 3 + 7 for x in cur[p]] for p in range(2500 - 100 * t)}, 3) for t in range(3)]))
    fbc = flatten(scs)
    for flags in (0, RELAX_TILES_64, 2):
        assert_same_outputs(fbc, _want16(fbc), emu_solve16(fbc, flags=flags), "emu 16-bit cells, Context in and out, flags %#x" % flags)
    # a Context whose counters leave the relaxation form's 16-bit fields: the scenario is left to the round form
    big = flatten([Scenario(brokers=list(range(40)), racks={b: "r%d" % (b % 8) for b in range(40)}, want_context=True,
                            context={b: {0: 65000, 1: 3} for b in range(40)},
                            topics=[Topic("t", {p: [(p + k) % 40 for k in range(3)] for p in range(900)}, 3)])])
    assert_same_outputs(big, _want16(big), emu_solve16(big), "emu 16-bit cells, Context beyond 16 bits: round form")
    # lists 5 wide: refused
    wide = _batch(5, 2, 600, 40, 10, 5, G.ACTIONS)
    with pytest.raises(RuntimeError, match="rc=-3"):
        emu_solve16(wide)
    # the sampled verification (round 6: instantiated for 16-bit cells too): same lists, never fires on an ascending LDS
    for flags in (VERIFY_SAMPLE(8), RELAX_TILES_64 | VERIFY_SAMPLE(255)):
        got = emu_solve16(fb, flags=flags)
        assert_same_outputs(fb, want, got, "emu 16-bit cells, sampled verification, flags %#x" % flags)
        assert_same_outputs(fbc, _want16(fbc), emu_solve16(fbc, flags=flags), "emu 16-bit cells, Context, sampled verification")


def test_emu16_descending_lane_order_is_caught_by_the_sampled_verification():
    """The 16-bit instances are what bench.py's cells16 leg and kas_solve_host16 launch: on an emulator whose LDS serves the lanes
    of one atomic in DESCENDING order they produce wrong lists with status OK, and with every tile verified none gets out."""
    from emu_lib import NO_RTN_QUOTA, RELAX_TILES_64, RELAX_TILES_128, VERIFY_SAMPLE, variant_solver16
    solve16 = variant_solver16("rtn_descending", ["-DKAS_EMU_RTN_DESCENDING"])
    fb = _batch(4321, 6, 6000, 80, 8, 3, ("remove1",))
    want = _want16(fb)
    ok = want.scenario_results["status"] == abi.KAS_OK
    assert ok.sum() >= 3
    P3 = 6000 * 3
    for tiles in (RELAX_TILES_64, RELAX_TILES_128):
        got = solve16(fb, NO_RTN_QUOTA | tiles)
        wrong = np.array([bool((want.out[s * P3:(s + 1) * P3] != got.out[s * P3:(s + 1) * P3]).any()) for s in range(6)]) & ok
        assert wrong.any(), "the descending LDS should have changed some list"
        assert (got.scenario_results["status"][wrong] == abi.KAS_OK).any()
        chk = solve16(fb, NO_RTN_QUOTA | tiles | VERIFY_SAMPLE(255))
        caught = chk.scenario_results["status"] == abi.KAS_FAIL_WATCHDOG
        assert caught[wrong].all(), "a scenario with a wrong list was not flagged"
        for s in np.nonzero(ok & ~caught)[0]:
            assert (want.out[s * P3:(s + 1) * P3] == chk.out[s * P3:(s + 1) * P3]).all()


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


# ---- the kernels' own 16-bit I/O on the GPU: kas_plan_create16 + kas_solve_device16 ----------------------------------
@pytest.mark.gpu
@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(scenarios())
def test_hip_device16_equals_oracle_small_odd_inputs(sc):
    from kafka_assigner_amd import native
    brokers, racks, topics = sc
    for want_ctx in (True, False):
        fb = flatten([Scenario(brokers=brokers, racks=racks, want_context=want_ctx,
                               topics=[Topic(n, c, rf, parts) for n, c, rf, parts in topics])])
        wide = bool((fb.topics["out_width"] > 3).any())
        try:
            got = native.solve_device16_with_flags(fb)
        except native.KasError as e:
            assert e.code == abi.KAS_E_UNSUPPORTED and wide, str(e)
            continue
        assert not wide
        assert_same_outputs(fb, _want16(fb), got, "hip, 16-bit cells in HBM, odd inputs")


@pytest.mark.gpu
@pytest.mark.parametrize("P,N,R,RF,actions", [
    (1000, 40, 8, 3, G.ACTIONS), (3000, 100, 10, 3, ("remove1",)), (2048, 64, 8, 2, ("add_k",)),
    (8000, 80, 8, 3, ("replace1", "add_k")),
])
def test_hip_device16_equals_oracle_seeded_batches_every_plan_variant(P, N, R, RF, actions):
    from kafka_assigner_amd import native
    fb = _batch(1234, 6, P, N, R, RF, actions)
    want = _want16(fb)
    got = native.solve_device16_with_flags(fb)
    assert "[16-bit cells]" in got.describe and "kas_order_relax_kernel" in got.describe
    assert_same_outputs(fb, want, got, "hip 16-bit cells")
    T64, T128 = abi.KAS_PLAN_RELAX_TILES_64, abi.KAS_PLAN_RELAX_TILES_128
    for flags, what in ((abi.KAS_PLAN_SPLIT_P4 | T64, "kas_p4_kernel + tiles of 64 rows: the headline's kernels"),
                        (abi.KAS_PLAN_FILL_WITH_P4, "first fit inside the fill workgroup"), (T128, "double tiles"),
                        (T64 | abi.KAS_PLAN_NO_RTN_QUOTA, "quota without the atomic-with-return"), (1, "general fill"),
                        (2, "round form"), (abi.KAS_PLAN_TICKET_ORDER, "ticket form asked for: round form"),
                        (T64 | (1 << 8), "one fill wavefront"), (8, "chunk-count pass")):
        assert_same_outputs(fb, want, native.solve_device16_with_flags(fb, flags), "hip 16-bit cells, " + what)
    # the sampled verification on the 16-bit instances (round 6): same lists, and it never fires on this hardware
    for flags in (abi.KAS_PLAN_VERIFY_SAMPLE(8), T64 | abi.KAS_PLAN_VERIFY_SAMPLE(255), T128 | abi.KAS_PLAN_VERIFY_SAMPLE(40)):
        gv = native.solve_device16_with_flags(fb, flags)
        assert "sampled verification" in gv.describe
        assert_same_outputs(fb, want, gv, "hip 16-bit cells, sampled verification, flags %#x" % flags)
    fb3 = _batch(78, 3, P, N, R, RF, ("remove1", "add_k"), cyclic=True)
    assert_same_outputs(fb3, _want16(fb3), native.solve_device16_with_flags(fb3), "hip 16-bit cells, cyclic rows")


@pytest.mark.gpu
def test_hip_device16_multi_topic_context_full_size_and_refusals():
    from kafka_assigner_amd import native
    from kafka_assigner_amd.flatten import node_set_batch
    fb = _multi_topic_scenarios(77, 3, 3, 700, 40, 8, 3)
    assert_same_outputs(fb, _want16(fb), native.solve_device16_with_flags(fb), "hip 16-bit cells, multi-topic")
    scs = []
    for s in range(3):
        cur = G.random_assignment(300 + s, 2500, 40, 8, 3)
        _, bs = G.scenario_action(300, s, 40, 8, actions=("remove1", "add_k"), max_add=4)
        racks = {int(b) * 3 + 7: "r%d" % int(r) for b, r in zip(bs.node_id, bs.node_rack)}
        scs.append(Scenario(brokers=[int(b) * 3 + 7 for b in bs.node_id], racks=racks, want_context=True,
                            topics=[Topic("topic-%d" % t, {p: [int(x) * 3 + 7 for x in cur[p]] for p in range(2500 - 100 * t)}, 3) for t in range(3)]))
    fbc = flatten(scs)
    assert_same_outputs(fbc, _want16(fbc), native.solve_device16_with_flags(fbc), "hip 16-bit cells, Context in and out")
    big = flatten([Scenario(brokers=list(range(40)), racks={b: "r%d" % (b % 8) for b in range(40)}, want_context=True,
                            context={b: {0: 65000, 1: 3} for b in range(40)},
                            topics=[Topic("t", {p: [(p + k) % 40 for k in range(3)] for p in range(900)}, 3)])])
    assert_same_outputs(big, _want16(big), native.solve_device16_with_flags(big), "hip 16-bit cells, Context beyond 16 bits: round form")
    # BASELINE configs[2]'s shape, 24 scenarios, every action of the bench mix; and 600 scenarios (first fit in kas_p4_kernel by size)
    S, P, N, R = 24, 100000, 1000, 10
    cur = np.stack([G.random_assignment(50 + s % 3, P, N, R, 3) for s in range(S)])
    sets = [G.scenario_action(9, s, N, R, actions=G.BENCH_ACTIONS, max_add=50)[1] for s in range(S)]
    full = node_set_batch([b.node_id for b in sets], [b.node_rack for b in sets], P, 3, 3, cur=cur)
    got = native.solve_device16_with_flags(full, abi.KAS_PLAN_SPLIT_P4 | abi.KAS_PLAN_RELAX_TILES_64)
    assert "kas_p4_kernel<3>" in got.describe and "tiles of 64 rows" in got.describe
    assert_same_outputs(full, oracle_16 := _want16(full), got, "hip 16-bit cells, 100k x 1k x RF 3")
    assert (oracle_16.scenario_results["status"] == abi.KAS_OK).sum() >= 20
    many = _batch(31, 600, 900, 60, 6, 3, G.BENCH_ACTIONS)
    gm = native.solve_device16_with_flags(many)
    assert "kas_p4_kernel<3> grid=600x64" in gm.describe
    assert_same_outputs(many, _want16(many), gm, "hip 16-bit cells, 600 scenarios")
    # refusals: lists 5 wide; an int32 plan through kas_solve_device16 and the other way round
    with pytest.raises(native.KasError) as e:
        native.solve_device16_with_flags(_batch(5, 2, 600, 40, 10, 5, G.ACTIONS))
    assert e.value.code == abi.KAS_E_UNSUPPORTED
    ctx = native.default_context()
    p32, p16 = native.Plan(ctx, fb), native.Plan(ctx, fb, cells16=True)
    p32.cells16, p16.cells16 = True, False                                # (call the wrong entry point on purpose)
    for pl in (p32, p16):
        with pytest.raises(native.KasError) as e:
            pl.solve_device(1, 1, 1, 1)
        assert e.value.code == abi.KAS_E_INVALID_ARG
        pl.close()


@pytest.mark.gpu
def test_hip16_wide_lists_do_not_thrash_the_plan_cache():
    """kas_solve_host16 on a batch the 16-bit kernels refuse (lists 5 wide: widened before and narrowed behind an
    int32 solve) used to ask the plan cache for a 16-bit plan first — rebuilding the int32 plan of the previous call in place,
    failing, destroying it — so that EVERY call rebuilt its plan.  The cell width of the solve is now decided from the shape
    before the cache is touched: the second call of the same batch finds its plan byte for byte and allocates nothing."""
    from kafka_assigner_amd import native
    ctx = native.DeviceContext(0)
    fb = _batch(5, 3, 900, 40, 10, 5, G.ACTIONS)
    want = oracle_solve(index_form(fb), threads=0)
    want.out = np.where(want.out < 0, 0xFFFF, want.out).astype(np.uint16)
    got = native.solve_host16(fb, ctx)
    np.testing.assert_array_equal(got.out[:fb.out_len], want.out[:fb.out_len])
    calls, hits, allocs = ctx.host_stats()
    for k in range(3):
        got = native.solve_host16(fb, ctx)
        np.testing.assert_array_equal(got.out[:fb.out_len], want.out[:fb.out_len])
        assert ctx.host_stats() == (calls + k + 1, hits + k + 1, allocs), ctx.host_stats()
    # ... and a batch the 16-bit kernels DO take, between two such calls, evicts nothing it needs
    fb3 = _batch(6, 3, 900, 40, 10, 3, G.ACTIONS)
    native.solve_host16(fb3, ctx)
    c2, h2, a2 = ctx.host_stats()
    native.solve_host16(fb, ctx); native.solve_host16(fb3, ctx)
    assert ctx.host_stats()[1] == h2 + 2 and ctx.host_stats()[2] == a2, ctx.host_stats()
    ctx.close()
