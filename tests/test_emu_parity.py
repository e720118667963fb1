"""Kernel-source parity on CPU: the solver body (kafka-assigner_amd/csrc/kas_solver_body.h),
compiled with g++ against the 64-fiber wave emulator of tests/emu, must reproduce the oracle
bit for bit.  This exercises the kernel's LOGIC and the product's planning code without a GPU;
the real parity tests (tests/test_hip_parity.py, -m gpu) run the same cases through the C ABI
on the MI355X.  The emulator is test infrastructure, never a product path."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings

from kafka_assigner_amd import abi
from kafka_assigner_amd.flatten import Scenario, Topic, flatten, uniform_batch
from kafka_assigner_amd import generator as G
from emu_lib import (FILL_WITH_P4, INDEX_ROWS, NO_INDEX_ROWS, NO_RTN_QUOTA, RELAX_TILES_64, RELAX_TILES_128, TICKET_ORDER, emu_solve, last_index_rows,
                     last_queue_rows, last_split_p4)
from oracle_lib import oracle_solve
from parity_util import assert_same_outputs
from test_oracle_vs_literal import scenarios


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(scenarios())
def test_emu_equals_oracle_small_odd_inputs(sc):
    brokers, racks, topics = sc
    fb = flatten([Scenario(brokers=brokers, racks=racks, want_context=True,
                           topics=[Topic(n, c, rf, parts) for n, c, rf, parts in topics])])
    want = oracle_solve(fb)
    assert_same_outputs(fb, want, emu_solve(fb), "emu")
    assert_same_outputs(fb, want, emu_solve(fb, flags=1), "emu generic fill")
    assert_same_outputs(fb, want, emu_solve(fb, flags=2 | (1 << 8)), "emu round order, 1 wave")
    assert_same_outputs(fb, want, emu_solve(fb, flags=(2 << 8) | (1 << 12)), "emu 2 waves, 1 scenario per wave")


@settings(max_examples=200, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(scenarios())
def test_emu_equals_oracle_small_odd_inputs_without_context_io(sc):
    """Same odd inputs, but no Context handed in or out: lists up to 3 wide then take the ticket form
    of P5 (ragged lists, duplicate brokers, partitions != keys(cur), empty topics, failures)."""
    brokers, racks, topics = sc
    fb = flatten([Scenario(brokers=brokers, racks=racks, want_context=False,
                           topics=[Topic(n, c, rf, parts) for n, c, rf, parts in topics])])
    want = oracle_solve(fb)
    assert_same_outputs(fb, want, emu_solve(fb), "emu (no ctx): relaxation form where applicable")
    # what the PRODUCT launches for a batch this small since round 6: first fit beside the order kernel in one workgroup where
    # that applies (kas_p4_order_kernel), index rows on request — ragged lists, duplicate brokers, empty topics, failures
    assert_same_outputs(fb, want, emu_solve(fb, p4_by_batch_size=True), "emu (no ctx), the product's choice for a small batch")
    assert_same_outputs(fb, want, emu_solve(fb, flags=0xC00000 | RELAX_TILES_64 | INDEX_ROWS), "emu (no ctx), first fit beside the order kernel, tiles of 64 rows, index rows")
    assert_same_outputs(fb, want, emu_solve(fb, flags=TICKET_ORDER), "emu (no ctx), ticket form")
    assert_same_outputs(fb, want, emu_solve(fb, flags=4 | (1 << 12)), "emu (no ctx), wide counters, 1 scenario per wave")


def _batch(seed, S, P, N, R, RF, actions, rack_aware=True, cyclic=False, name_hash=3644):
    curs, ids, racks = [], [], []
    nmax = 0
    sets = []
    for s in range(S):
        cur = G.cyclic_assignment(P, N, RF, s) if cyclic else G.random_assignment(seed + s, P, N, R, RF)
        act, bs = G.scenario_action(seed, s, N, R, actions=actions, max_add=max(2, N // 10))
        if not rack_aware:
            bs = G.BrokerSet(bs.node_id, np.arange(bs.node_id.shape[0], dtype=np.int32))
        curs.append(cur); sets.append(bs)
    # uniform_batch needs equal N per scenario: build the descriptors per scenario instead
    from kafka_assigner_amd.flatten import FlatBatch
    scen = np.zeros(S, dtype=abi.SCENARIO_DESC_DTYPE)
    topics = np.zeros(S, dtype=abi.TOPIC_DESC_DTYPE)
    node_id = np.concatenate([b.node_id for b in sets]).astype(np.int32)
    node_rack = np.concatenate([b.node_rack for b in sets]).astype(np.int32)
    off = 0
    for s, b in enumerate(sets):
        scen[s] = (b.node_id.shape[0], s, 1, 0, off, -1)
        off += b.node_id.shape[0]
        topics[s] = (name_hash, P, RF, RF, RF, 0, s * P * RF, s * P * RF, -1, -1, -1)
    return FlatBatch(scen=scen, topics=topics, node_id=node_id, node_rack=node_rack,
                     cur=np.concatenate([c.reshape(-1) for c in curs]).astype(np.int32),
                     aux=np.zeros(0, np.int32), ctx=np.zeros(0, np.int32), out_len=S * P * RF)


@pytest.mark.parametrize("P,N,R,RF,actions", [
    (1000, 40, 8, 3, G.ACTIONS),            # many tiles, every action kind
    (3000, 100, 10, 3, ("remove1",)),       # C2-shaped, scaled down
    (2048, 64, 8, 2, ("add_k",)),           # N power of two, RF 2
    (777, 40, 10, 5, G.ACTIONS),            # RF 5, ragged last tile
    (640, 24, 8, 4, ("replace1", "remove1")),
    (8000, 80, 8, 3, ("replace1", "add_k")),  # many P4 windows in flight on all waves; zero-slack strandings
])
def test_emu_equals_oracle_seeded_batches(P, N, R, RF, actions):
    fb = _batch(1234, 6, P, N, R, RF, actions)
    want = oracle_solve(fb)
    assert_same_outputs(fb, want, emu_solve(fb, flags=INDEX_ROWS), "emu, index rows")
    # round 6: lists up to 3 wide — the fill's first scan leaves every row's node indices where the mid rows go, the second streams those
    assert last_index_rows() == (6 if RF <= 3 else 0)
    assert_same_outputs(fb, want, emu_solve(fb, flags=INDEX_ROWS | FILL_WITH_P4 | RELAX_TILES_64), "emu, index rows, first fit inside the fill workgroup, tiles of 64 rows")
    assert_same_outputs(fb, want, emu_solve(fb, flags=NO_INDEX_ROWS), "emu, cur read by both row scans of the fill (no index rows)")
    assert last_index_rows() == 0
    assert_same_outputs(fb, want, emu_solve(fb, flags=NO_INDEX_ROWS | FILL_WITH_P4 | RELAX_TILES_64), "emu, no index rows, first fit inside the fill workgroup, tiles of 64 rows")
    assert_same_outputs(fb, want, emu_solve(fb), "emu")
    assert last_split_p4() == 1                             # round 5: first fit in kas_p4_kernel behind the fill kernel ...
    assert_same_outputs(fb, want, emu_solve(fb, flags=FILL_WITH_P4), "emu, first fit inside the fill workgroup")
    assert last_split_p4() == 0                             # ... unless the plan says otherwise ...
    assert_same_outputs(fb, want, emu_solve(fb, p4_by_batch_size=True), "emu, first fit where the product runs it for six scenarios")
    assert last_split_p4() == 0                             # ... or the batch is small (kas_split_p4: from 512 scenarios on)
    assert_same_outputs(fb, want, emu_solve(fb, flags=FILL_WITH_P4 | RELAX_TILES_64), "emu, first fit inside the fill workgroup, tiles of 64 rows")
    assert_same_outputs(fb, want, emu_solve(fb, flags=TICKET_ORDER), "emu ticket form")
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_64), "emu relaxation form, tiles of 64 rows")
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_64 | NO_RTN_QUOTA), "emu tiles of 64 rows, quota drawn without the atomic-with-return")
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_64 | (1 << 8)), "emu tiles of 64 rows, one fill wavefront per scenario")
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_64 | 8), "emu tiles of 64 rows, chunk-count pass")
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_128), "emu relaxation form, double tiles")
    assert_same_outputs(fb, want, emu_solve(fb, flags=NO_RTN_QUOTA), "emu fill, quota drawn without the atomic-with-return")
    assert_same_outputs(fb, want, emu_solve(fb, flags=1), "emu generic fill")
    # every workgroup width, and the round form of the preference ordering
    for nw, g in ((1, 1), (2, 2), (8, 4)):
        assert_same_outputs(fb, want, emu_solve(fb, flags=(nw << 8) | (g << 12)), f"emu {nw} waves, {g} scenarios per wave")
    assert_same_outputs(fb, want, emu_solve(fb, flags=2), "emu round order")
    assert_same_outputs(fb, want, emu_solve(fb, flags=4), "emu 4 x uint16 counter rows")
    assert_same_outputs(fb, want, emu_solve(fb, flags=3 | (1 << 8)), "emu generic fill + round order, 1 wave")
    # the generator must produce solvable scenarios most of the time, else the test is vacuous
    assert (want.scenario_results["status"] == abi.KAS_OK).sum() >= 1


def test_emu_rack_awareness_disabled_and_cyclic_failure():
    # --disable_rack_awareness: every broker its own rack
    fb = _batch(99, 4, 1500, 50, 10, 3, G.ACTIONS, rack_aware=False)
    assert_same_outputs(fb, oracle_solve(fb), emu_solve(fb), "emu norack")
    # perfectly cyclic start + added brokers strands partitions in the reference (Q9): the
    # failing partition id must match
    fb = _batch(7, 4, 1200, 60, 6, 3, ("add_k", "remove1"), cyclic=True)
    want = oracle_solve(fb)
    assert_same_outputs(fb, want, emu_solve(fb), "emu cyclic")


def test_emu_sparse_broker_ids_use_binary_search():
    cur = G.random_assignment(5, 500, 20, 5, 3).astype(np.int64) * 100003 + 7   # sparse ids
    ids = (np.arange(20, dtype=np.int64) * 100003 + 7).astype(np.int32)[None, :]
    racks = (np.arange(20) % 5).astype(np.int32)[None, :]
    fb = uniform_batch(cur.astype(np.int32)[None], ids[:, :19], racks[:, :19], 3)
    assert_same_outputs(fb, oracle_solve(fb), emu_solve(fb), "emu sparse")


def test_emu_config5_broker_count_uses_general_fill_without_histogram_lds():
    """BASELINE.json configs[4] shape: 5k brokers x 40 racks, RF 5, mixed remove + add.  The
    histogram/quota table of the rack-diverse fill does not fit 160 KiB of LDS at this size, so
    the plan falls back to the general sticky fill (and the round form of P5: lists are 5 wide)."""
    P, N, R, RF = 3000, 5000, 40, 5
    cur = G.random_assignment(7, P, N, R, RF)
    bs = G.perturb_brokers(N, R, remove=list(range(0, N, 50)), add=200)
    fb = uniform_batch(cur[None], bs.node_id[None], bs.node_rack[None], RF)
    assert_same_outputs(fb, oracle_solve(fb), emu_solve(fb), "emu C5 shape")


def _multi_topic_scenarios(seed, n_scen, n_topics, P, N, R, RF):
    """Scenarios of several topics sharing one Context that is NOT handed in or out: the shape of one
    PRINT_REASSIGNMENT run (KAG:172-184) — and the ticket form's cross-topic case (a topic's
    tickets start where the previous topic's left off)."""
    scs = []
    for s in range(n_scen):
        act, bs = G.scenario_action(seed, s, N, R, actions=G.BENCH_ACTIONS, max_add=max(2, N // 10))
        racks = {int(b): "r%d" % int(r) for b, r in zip(bs.node_id, bs.node_rack)}
        topics = []
        for t in range(n_topics):
            cur = G.random_assignment(seed + 31 * s + t, P + 17 * t, N, R, RF)
            topics.append(Topic("topic-%d" % t, {p: cur[p].tolist() for p in range(cur.shape[0])}, RF))
        scs.append(Scenario(brokers=[int(b) for b in bs.node_id], racks=racks, topics=topics))
    return flatten(scs)


def test_emu_multi_topic_scenarios_without_context_io_use_cross_topic_tickets():
    fb = _multi_topic_scenarios(77, 3, 3, 700, 40, 8, 3)
    assert (fb.scen["ctx_off"] < 0).all() and (fb.scen["topic_count"] == 3).all()
    want = oracle_solve(fb)
    assert (want.topic_results["status"] == abi.KAS_OK).sum() >= 4      # and some that fail or are skipped
    assert_same_outputs(fb, want, emu_solve(fb), "emu multi-topic, relaxation form")
    assert_same_outputs(fb, want, emu_solve(fb, flags=TICKET_ORDER), "emu multi-topic tickets")
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_64), "emu multi-topic, relaxation form over tiles of 64 rows")
    assert_same_outputs(fb, want, emu_solve(fb, flags=2), "emu multi-topic rounds")
    assert_same_outputs(fb, want, emu_solve(fb, flags=(2 << 12) | (2 << 8)), "emu multi-topic, 2 scenarios per wave")


def test_emu_queue_path_runs_and_agrees():
    """Added brokers fill up from consecutive orphans, so rows wait in line on one node: the ticket
    form decides such queues in one step (thresholds + prefix sums in rank space).  Assert that
    path really ran here (the emulator counts the rows decided inside queues) and agrees with the
    oracle, for both group widths and both counter layouts."""
    fb = _batch(4321, 4, 6000, 100, 10, 3, ("add_k", "mixed"))
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_OK).any()
    for flags in (TICKET_ORDER, 1 << 12, 4 | (4 << 12)):
        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu flags {flags:#x}")
        assert last_queue_rows() > 20, "the queue path did not run"
    assert_same_outputs(fb, want, emu_solve(fb), "emu relaxation form")


def test_emu_protocols_survive_arbitrary_wave_speeds():
    """The waves of a workgroup talk through LDS words only (ring tags of the order kernel, P4
    progress words of the fill kernel).  KAS_EMU_CHAOS makes the emulator release waves at random
    and hold one back for long stretches; the results must not change.  (Own processes: the seed
    is read once per process.)"""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')\n"
        "from test_emu_parity import _batch\n"
        "from emu_lib import emu_solve\n"
        "from oracle_lib import oracle_solve\n"
        "from parity_util import assert_same_outputs\n"
        "from kafka_assigner_amd import generator as G\n"
        "for acts, P, N in ((G.ACTIONS, 2000, 60), (('replace1', 'add_k'), 3500, 80)):\n"
        "    fb = _batch(99, 4, P, N, 8, 3, acts)\n"
        "    want = oracle_solve(fb)\n"
        "    for flags in (0, 0x10000, 1 << 12, (8 << 8) | (4 << 12), 0xC00000, 0xC00000 | 0x20000):\n"   # 0xC00000: first fit as a wavefront of the order kernel's workgroup
        "        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), 'chaos')\n"
        "for rf, acts in ((5, ('add_k', 'mixed')), (4, G.ACTIONS)):\n"      # the wide ticket form: joint solve, claim lists
        "    fb = _batch(77, 2, 1500, 120, 12, rf, acts)\n"
        "    assert_same_outputs(fb, oracle_solve(fb), emu_solve(fb), 'chaos, wide lists')\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for seed in ("1", "7"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                           env=dict(os.environ, KAS_EMU_CHAOS=seed), timeout=900)
        assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_emu_topic_without_rows_next_to_full_width_topics():
    """A topic with zero partitions whose widths match the kernel's width class: the fast fill's
    full-row loads re-read the last row for lanes past the end, and there is no row."""
    cur = G.cyclic_assignment(300, 12, 3)                # balanced and rack-diverse: topic a succeeds
    sc = Scenario(brokers=list(range(12)), racks={b: "r%d" % (b % 4) for b in range(12)},
                  topics=[Topic("a", {p: cur[p].tolist() for p in range(300)}, 3), Topic("empty", {}, 3)])
    fb = flatten([sc])
    fb.topics["cur_width"][1] = 3; fb.topics["out_width"][1] = 3
    want = oracle_solve(fb)
    assert want.topic_results["status"][1] == abi.KAS_OK
    assert_same_outputs(fb, want, emu_solve(fb), "emu empty topic last")


@pytest.mark.parametrize("P,N,R,RF,actions,rack_aware", [
    (3200, 120, 12, 5, ("add_k",), False),    # one broker at a time fills up: long single-node queues
    (2600, 120, 12, 4, ("mixed",), True),     # rack constraints interleave several nodes being filled
    (2000, 120, 12, 5, G.ACTIONS, True),
    (777, 40, 10, 4, G.ACTIONS, True),
])
def test_emu_wide_lists_take_the_wide_ticket_form(P, N, R, RF, actions, rack_aware):
    """Lists 4 and 5 wide without Context in/out: kas_order_wide.h (tickets, five 10-bit counts per
    node, the generalised queue step) against the oracle and against the round form."""
    fb = _batch(100, 2, P, N, R, RF, actions, rack_aware=rack_aware)
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_OK).any()
    assert_same_outputs(fb, want, emu_solve(fb), "emu wide tickets")
    if P >= 3000:
        assert last_queue_rows() > 0, "the queue path of the wide kernel did not run"
    assert_same_outputs(fb, want, emu_solve(fb, flags=2), "emu round form")
    assert_same_outputs(fb, want, emu_solve(fb, flags=1 | (2 << 8)), "emu wide tickets after the general fill, 2 waves")


def test_emu_wide_lists_multi_topic_and_mixed_widths():
    """Topics of different widths in one scenario (3-, 5- and 4-wide lists: the batch runs at width
    class 5), tickets carried across topics."""
    scs = []
    for s in range(3):
        act, bs = G.scenario_action(5, s, 60, 12, actions=("add_k",), max_add=6)
        racks = {int(b): "r%d" % int(r) for b, r in zip(bs.node_id, bs.node_rack)}
        topics = []
        for t, rf in enumerate((3, 5, 4)):
            cur = G.random_assignment(9 + 7 * s + t, 900 + 31 * t, 60, 12, rf)
            topics.append(Topic("topic-%d" % t, {p: cur[p].tolist() for p in range(cur.shape[0])}, rf))
        scs.append(Scenario(brokers=[int(b) for b in bs.node_id], racks=racks, topics=topics))
    fb = flatten(scs)
    want = oracle_solve(fb)
    assert (want.topic_results["status"][3:6] == abi.KAS_OK).all()      # one scenario runs all three widths
    assert_same_outputs(fb, want, emu_solve(fb), "emu wide multi-topic")
    assert_same_outputs(fb, want, emu_solve(fb, flags=2), "emu wide multi-topic, round form")


def test_emu_fill_with_per_chunk_histograms_and_with_the_chunk_count_pass():
    """The rack-diverse fill counts its sweep histograms per chunk in the first pass over cur when the
    LDS allows (lists <= 3 wide): no second counting pass.  Both forms against the oracle, at every
    workgroup width that has chunks, with quota edges inside tiles (small clusters) and sparse ids."""
    from emu_lib import last_fused
    for P, N, R, RF, acts in ((9000, 90, 9, 3, G.ACTIONS), (5000, 40, 8, 2, ("add_k", "remove1")),
                              (4000, 33, 11, 3, ("mixed", "replace1"))):
        fb = _batch(555, 4, P, N, R, RF, acts)
        want = oracle_solve(fb)
        assert (want.scenario_results["status"] == abi.KAS_OK).any()
        for nw in (0, 2, 8):
            assert_same_outputs(fb, want, emu_solve(fb, flags=nw << 8), f"emu per-chunk histograms, waves {nw}")
            assert last_fused()
            assert_same_outputs(fb, want, emu_solve(fb, flags=8 | (nw << 8)), f"emu chunk-count pass, waves {nw}")
            assert not last_fused()
        assert_same_outputs(fb, want, emu_solve(fb, flags=1 << 8), "emu one wave: nothing to fuse")
        assert not last_fused()
    cur = G.random_assignment(5, 2500, 20, 5, 3).astype(np.int64) * 100003 + 7            # sparse ids: binary search
    ids = (np.arange(20, dtype=np.int64) * 100003 + 7).astype(np.int32)[None, :]
    racks = (np.arange(20) % 5).astype(np.int32)[None, :]
    fb = uniform_batch(cur.astype(np.int32)[None], ids[:, :19], racks[:, :19], 3)
    assert_same_outputs(fb, oracle_solve(fb), emu_solve(fb), "emu per-chunk histograms, sparse ids")
    assert last_fused()


def test_emu_index_rows_where_they_apply_and_where_they_do_not():
    """KAS_FLAG_INDEX_ROWS (round 6): int32 cells, per-chunk histograms, the quota drawn with the atomic-with-return, a direct id
    table and rows of the batch's width — the fill's first scan stores every row's node indices (0xffff: the broker left the set)
    where its mid row goes and the second scan streams those, storing only the rows that do not keep all their replicas.  Anything
    else reads `cur` twice as before; both against the oracle."""
    # duplicates in a row, brokers that are not in the set, ids far from zero (min_id != 0), ragged last tile
    rng = np.random.default_rng(61)
    P, N = 1333, 37
    ids = (np.arange(N, dtype=np.int32) * 3 + 1000)[None, :]
    racks = (np.arange(N) % 7).astype(np.int32)[None, :]
    cur = (rng.integers(0, N + 6, size=(P, 3)).astype(np.int32) * 3 + 1000)      # (some ids beyond the set, some rows with one broker twice)
    cur[::17, 1] = cur[::17, 0]
    fb = uniform_batch(cur[None], ids, racks, 3)
    want = oracle_solve(fb)
    assert_same_outputs(fb, want, emu_solve(fb, flags=INDEX_ROWS), "emu index rows: duplicates, absent brokers (rows not rack-diverse: the general fill over the scratch)")
    cur2 = G.random_assignment(8, P, N, 7, 3) * 3 + 1000
    cur2[5::11, 2] = 1000 + 3 * (N + 2)                                          # a broker that left the set, in rack-diverse rows
    fb2 = uniform_batch(cur2.astype(np.int32)[None], ids[:, :N - 2], racks[:, :N - 2], 3)
    want2 = oracle_solve(fb2)
    assert (want2.scenario_results["status"] == abi.KAS_OK).all()
    assert_same_outputs(fb2, want2, emu_solve(fb2, flags=INDEX_ROWS), "emu index rows: absent brokers, ids from 1000")
    assert last_index_rows() == 1
    assert_same_outputs(fb2, want2, emu_solve(fb2, flags=NO_INDEX_ROWS), "emu without index rows")
    for flags, n in ((NO_RTN_QUOTA, 0), (8, 0), (1 << 8, 0), (1, 0), (2 << 8, 1), (8 << 8, 1), (FILL_WITH_P4, 1), (2, 1), (TICKET_ORDER, 1)):
        assert_same_outputs(fb2, want2, emu_solve(fb2, flags=flags | INDEX_ROWS), f"emu index rows, plan flags {flags:#x}")
        assert last_index_rows() == n, (flags, last_index_rows())
    # topics of a scenario decide one by one: a topic narrower than the batch, a topic growing its lists (cur 2 wide, out 3 wide),
    # a topic with a partition subset (in_partitions) — rows of the batch's width take the index rows, the others do not
    scs = [Scenario(brokers=list(range(30)), racks={b: "r%d" % (b % 6) for b in range(30)}, want_context=False,
                    topics=[Topic("a", {p: G.random_assignment(1, 700, 30, 6, 3)[p].tolist() for p in range(700)}, 3),
                            Topic("b", {p: G.random_assignment(2, 500, 30, 6, 2)[p].tolist() for p in range(500)}, 3),
                            Topic("c", {p: G.random_assignment(3, 300, 30, 6, 2)[p].tolist() for p in range(300)}, 2),
                            Topic("d", {p: G.random_assignment(4, 900, 30, 6, 3)[p].tolist() for p in range(900)}, 3)])]
    fbm = flatten(scs)
    wantm = oracle_solve(fbm)
    assert_same_outputs(fbm, wantm, emu_solve(fbm, flags=INDEX_ROWS), "emu index rows, topics of several widths")
    assert last_index_rows() == 2                                                # topics a and d
    # sparse ids (binary search): no direct table, no index rows
    curs = G.random_assignment(5, 2500, 20, 5, 3).astype(np.int64) * 100003 + 7
    sid = (np.arange(20, dtype=np.int64) * 100003 + 7).astype(np.int32)[None, :]
    fbs = uniform_batch(curs.astype(np.int32)[None], sid[:, :19], (np.arange(20) % 5).astype(np.int32)[None, :19], 3)
    assert_same_outputs(fbs, oracle_solve(fbs), emu_solve(fbs, flags=INDEX_ROWS), "emu sparse ids")
    assert last_index_rows() == 0


def test_emu_slim_fill_kernel_in_front_and_the_full_kernel_for_what_it_hands_back():
    """kas_fill_slim_kernel (round 6): int32 cells, lists up to 3 wide, per-chunk histograms, a direct id table, first fit handed
    over — the slim kernel takes every scenario, solves the rack-diverse ones on the one path it holds and flags the others
    (rows not rack-diverse, a topic without rows, ...), which kas_fill_kernel then solves from their first topic; topics narrower
    than the batch stay with the slim kernel (its scans take ragged rows)."""
    from emu_lib import FULL_FILL, P4_WITH_ORDER, last_slim_fill
    # rack-diverse scenarios: all of them stay with the slim kernel, under either hand-over of first fit
    fb = _batch(1234, 5, 1777, 45, 9, 3, G.BENCH_ACTIONS)
    want = oracle_solve(fb)
    for flags in (0, P4_WITH_ORDER, RELAX_TILES_128, TICKET_ORDER, 2):
        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu slim fill, plan flags {flags:#x}")
        assert last_slim_fill() == 5, (flags, last_slim_fill())
    for flags in (FULL_FILL, INDEX_ROWS, NO_RTN_QUOTA, 8, 1, FILL_WITH_P4, 2 << 8):   # switched off / forms it does not hold
        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu without the slim fill, plan flags {flags:#x}")
        assert last_slim_fill() == 0, (flags, last_slim_fill())
    fb2 = _batch(77, 3, 900, 30, 6, 2, G.ACTIONS)
    assert_same_outputs(fb2, oracle_solve(fb2), emu_solve(fb2), "emu slim fill, lists 2 wide")
    assert last_slim_fill() == 3
    # scenarios 1 and 3 start from rows that are not rack-diverse (cyclic over 6 racks that divide the broker count): handed back
    racks = (np.arange(60) % 6).astype(np.int32)
    ids = np.arange(60, dtype=np.int32)
    curs = [G.random_assignment(5, 1200, 60, 6, 3), G.cyclic_assignment(1200, 60, 3, 1) * 6 % 60,
            G.random_assignment(6, 1200, 60, 6, 3), G.cyclic_assignment(1200, 60, 3, 2) * 6 % 60]
    fbm = uniform_batch(np.stack(curs).astype(np.int32), np.tile(ids, (4, 1))[:, :58], np.tile(racks, (4, 1))[:, :58], 3)
    wantm = oracle_solve(fbm)
    assert_same_outputs(fbm, wantm, emu_solve(fbm), "emu slim fill: two of four scenarios handed back")
    assert last_slim_fill() == 2
    assert_same_outputs(fbm, wantm, emu_solve(fbm, flags=P4_WITH_ORDER), "emu slim fill + first fit beside the order kernel: two of four handed back")
    assert last_slim_fill() == 2
    # scenarios of several topics, the third narrower than the batch (2 wide in a batch 3 wide): the slim kernel's scans take the
    # ragged rows themselves — nothing is handed back
    scs = [Scenario(brokers=list(range(30)), racks={b: "r%d" % (b % 6) for b in range(30)}, want_context=False,
                    topics=[Topic("a", {p: G.random_assignment(1, 700, 32, 6, 3)[p].tolist() for p in range(700)}, 3),
                            Topic("d", {p: G.random_assignment(4, 900, 32, 6, 3)[p].tolist() for p in range(900)}, 3),
                            Topic("c", {p: G.random_assignment(3, 300, 32, 6, 2)[p].tolist() for p in range(300)}, 2)]),
           Scenario(brokers=list(range(31)), racks={b: "r%d" % (b % 6) for b in range(31)}, want_context=False,
                    topics=[Topic("a", {p: G.random_assignment(9, 800, 32, 6, 3)[p].tolist() for p in range(800)}, 3)])]
    fbt = flatten(scs)
    wantt = oracle_solve(fbt)
    got = emu_solve(fbt)
    assert_same_outputs(fbt, wantt, got, "emu slim fill: a topic narrower than the batch")
    assert last_slim_fill() == 2
    assert_same_outputs(fbt, wantt, emu_solve(fbt, flags=FULL_FILL), "emu: the same without the slim kernel")
    # scenarios whose SECOND topic starts from rows that are not rack-diverse: the slim kernel has solved topic one by then (records,
    # mid rows, hand-over words) — the full kernel solves the scenario again from its first topic over all of that
    fb3 = _later_topic_hands_back()
    want3 = oracle_solve(fb3)
    assert sorted(set(want3.scenario_results["status"].tolist())) == [abi.KAS_OK, abi.KAS_FAIL_UNASSIGNABLE]   # (some of them strand, KAS:183-184)
    for flags in (0, P4_WITH_ORDER):
        assert_same_outputs(fb3, want3, emu_solve(fb3, flags=flags), f"emu slim fill: the second topic hands the scenario back, plan flags {flags:#x}")
        assert last_slim_fill() == 4


def test_emu_fill_kernel_deals_the_scenarios_handed_back_by_rank(monkeypatch):
    """kas_fill_kernel's own loop (kas::fill_block, round 6): a launch for flagged scenarios only deals them to its workgroups by
    RANK among the flagged ones and workgroup 0 leaves their number in KasLaunch::handback (the plan sizes its next such launch by it);
    a launch for every scenario deals them by index.  Nine scenarios of which five (1, 2, 5, 7, 8) start from rows that are not
    rack-diverse, on grids of 1, 2, 3, 4 and 9 workgroups (KAS_EMU_FILL_GRID: the emulator runs the workgroups one after another,
    each over the scenarios it takes, its LDS not cleared between them) — lists equal to the oracle's, five counted; the same
    batches with the slim kernel off (every scenario by index on those grids), and a batch nothing of which goes back (count 0)."""
    from emu_lib import FULL_FILL, P4_WITH_ORDER, last_handback, last_slim_fill
    S, P, N = 9, 500, 60
    racks = (np.arange(N) % 6).astype(np.int32)
    ids = np.arange(N, dtype=np.int32)
    back = (1, 2, 5, 7, 8)
    curs = [(G.cyclic_assignment(P, N, 3, s) * 6 % N) if s in back else G.random_assignment(300 + s, P, N, 6, 3) for s in range(S)]
    fb = uniform_batch(np.stack(curs).astype(np.int32), np.tile(ids, (S, 1))[:, :58], np.tile(racks, (S, 1))[:, :58], 3)
    want = oracle_solve(fb)
    fb0 = _batch(4321, 7, 600, 45, 9, 3, G.BENCH_ACTIONS)               # rack-diverse: nothing handed back
    want0 = oracle_solve(fb0)
    for grid in (1, 2, 3, 4, 9):
        monkeypatch.setenv("KAS_EMU_FILL_GRID", str(grid))
        for flags in (0, P4_WITH_ORDER):
            assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu: five of nine handed back, {grid} workgroups behind the slim kernel, plan flags {flags:#x}")
            assert last_slim_fill() == S - len(back) and last_handback() == len(back), (grid, flags, last_slim_fill(), last_handback())
        assert_same_outputs(fb, want, emu_solve(fb, flags=FULL_FILL), f"emu: every scenario by index on {grid} workgroups")
        assert last_slim_fill() == 0 and last_handback() == -1
        assert_same_outputs(fb0, want0, emu_solve(fb0), f"emu: nothing handed back, {grid} workgroups find nothing to do")
        assert last_slim_fill() == 7 and last_handback() == 0


def _later_topic_hands_back():
    """8 scenarios of three topics over 48-50 brokers in 10 racks; in every other one the SECOND topic's rows are not rack-diverse"""
    scs = []
    for s in range(8):
        n = 50 - (s % 3)
        cyc = G.cyclic_assignment(701, 50, 3, s) * 10 % 50
        second = Topic("b", {p: (cyc[p] if s % 2 == 0 else G.random_assignment(20 + s, 701, 50, 10, 3)[p]).tolist() for p in range(701)}, 3)
        scs.append(Scenario(brokers=list(range(n)), racks={b: "r%d" % (b % 10) for b in range(n)}, want_context=False,
                            topics=[Topic("a", {p: G.random_assignment(11 + s, 901, 50, 10, 3)[p].tolist() for p in range(901)}, 3), second,
                                    Topic("c", {p: G.random_assignment(13 + s, 503, 50, 10, 3)[p].tolist() for p in range(503)}, 3)]))
    return flatten(scs)


def test_emu_relaxation_form_broker_ids_from_the_lds_and_from_the_node_table(monkeypatch):
    """Round 6: the relaxation form's instances for int32 cells keep the scenario's broker ids in the LDS (kas_relax_lds_ids) and read a
    final row's ids there; broker counts whose ids do not fit keep the gather from the node table (asked for with the asynchronous
    loads, waited for at the step's one s_waitcnt).  Both against the oracle, with and without a Context, both tile sizes."""
    from emu_lib import last_relax_idl
    fb = _batch(606, 5, 5000, 90, 9, 3, G.ACTIONS)
    fb.node_id[:] = fb.node_id * 5 + 77                                      # ids that are not their own index
    fb.cur[:] = np.where(fb.cur >= 0, fb.cur * 5 + 77, fb.cur)
    want = oracle_solve(fb)
    fbm = _multi_topic_scenarios(78, 3, 3, 900, 40, 8, 3)
    wantm = oracle_solve(fbm)
    for gather in ("0", "1"):
        monkeypatch.setenv("KAS_EMU_RELAX_GATHER", gather)
        for flags in (RELAX_TILES_64, RELAX_TILES_128):
            assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu relaxation form, gather={gather}, flags {flags:#x}")
            assert last_relax_idl() == (0 if gather == "1" else 1)
            assert_same_outputs(fbm, wantm, emu_solve(fbm, flags=flags), f"emu relaxation form, topics + Context, gather={gather}, flags {flags:#x}")
    monkeypatch.setenv("KAS_EMU_RELAX_GATHER", "0")
    # 9,000 brokers: the ids no longer fit beside the counter words (KAS_RELAX_IDS_LDS_MAX) — the gather instances by themselves
    big = _batch(17, 1, 30000, 9000, 10, 3, ("remove1",))
    assert_same_outputs(big, oracle_solve(big), emu_solve(big), "emu relaxation form, 9,000 brokers")
    assert last_relax_idl() == 0


@pytest.mark.parametrize("S,P,N,R,RF,actions,rack_aware", [
    (6, 777, 40, 10, 5, G.ACTIONS, True),              # ragged last tile, every action kind
    (6, 640, 24, 8, 4, ("replace1", "remove1"), True),
    (4, 3000, 120, 24, 5, ("add_k",), True),
    (3, 5000, 100, 10, 4, G.ACTIONS, False),           # rack awareness off: every broker its own rack
])
def test_emu_wide_lists_relaxation_form(S, P, N, R, RF, actions, rack_aware):
    """Round 6: the relaxation form for lists 4 and 5 wide (kas_order_relax_wide.h; KAS_PLAN_RELAX_TILES(1) at these widths) —
    uint64 counter words of four 16-bit fields, sorted holders, the general pick over the holders that are left — against the
    oracle; without the flag the wide ticket form runs."""
    from emu_lib import last_order_form, last_relax_stats
    fb = _batch(1234, S, P, N, R, RF, actions, rack_aware=rack_aware)
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_OK).any()
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_64), "emu relaxation form, wide lists")
    assert last_order_form() == 4
    tiles, evals, _ = last_relax_stats()
    assert tiles > 0 and evals >= 2 * tiles                 # (every tile: one evaluation to decide, one to see nothing moved)
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_64 | FILL_WITH_P4), "emu relaxation form, wide lists, first fit inside the fill workgroup")
    assert_same_outputs(fb, want, emu_solve(fb), "emu wide ticket form")
    assert last_order_form() == 2
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_64 | TICKET_ORDER), "emu: the ticket form asked for by name")
    assert last_order_form() == 2


def test_emu_wide_lists_relaxation_form_topics_of_several_widths_share_the_counters():
    """A topic 5 wide, one 4 wide, one 3 wide and one 5 wide again in ONE scenario: every list position r <= 3 adds to its field
    whatever the list's length (KAS:236), and a later, wider topic reads what the narrower one left (KAS:263-278)."""
    from emu_lib import last_order_form
    scs = []
    for s in range(3):
        brokers = list(range(50)) if s == 0 else [b for b in range(53) if b != 7 * s]
        tps = []
        for t, (w, P) in enumerate(((5, 900), (4, 700), (3, 500), (5, 300), (2, 200))):
            cur = G.random_assignment(50 + s * 7 + t, P, 50, 10, w)
            tps.append(Topic("topic-%d" % t, {p: cur[p].tolist() for p in range(P)}, w))
        # (no rack map — every broker its own rack, KAS:82-86: with racks of equal size a load cap of ceil(mean) leaves the rack-aware
        # first fit no slack at these widths, and a failed topic would end the scenario before the widths have met)
        scs.append(Scenario(brokers=brokers, racks={}, want_context=False, topics=tps))
    fb = flatten(scs)
    want = oracle_solve(fb)
    assert (want.topic_results["status"] == abi.KAS_OK).sum() >= 10, want.topic_results["status"]
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_64), "emu relaxation form, wide lists, topics of several widths")
    assert last_order_form() == 4
    # with a Context handed in the form does not apply: the wide ticket form + round form as before
    for sc in scs:
        sc.want_context = True
    fbc = flatten(scs)
    assert_same_outputs(fbc, oracle_solve(fbc), emu_solve(fbc, flags=RELAX_TILES_64), "emu wide lists with a Context: ticket form")
    assert last_order_form() == 2


@pytest.mark.parametrize("P,N,R,RF,actions", [
    (1000, 40, 8, 3, G.ACTIONS),
    (3000, 100, 10, 3, ("remove1",)),
    (2048, 64, 8, 2, ("add_k",)),
    (8000, 80, 8, 3, ("replace1", "add_k")),   # most scenarios strand a partition (KAS:183-184): the abandon path
])
def test_emu_first_fit_as_a_wavefront_of_the_order_kernels_workgroup(P, N, R, RF, actions):
    """Round 6, kas_p4_order_kernel (KAS_PLAN_SPLIT_P4 | KAS_PLAN_FILL_WITH_P4): first fit runs as the second wavefront of the
    relaxation form's workgroup and the order wavefront follows its progress — it asks for a tile's mid rows only when first fit is
    done with them, abandons a topic first fit fails (whose rows the first-fit wavefront then pads) and writes its digest behind
    first fit's records.  Against the oracle, both tile sizes, both cell widths, with and without index rows."""
    from emu_lib import INDEX_ROWS, P4_WITH_ORDER, emu_solve16, last_p4_order
    from test_cells16 import _want16
    fb = _batch(1234, 6, P, N, R, RF, actions)
    want = oracle_solve(fb)
    for flags in (P4_WITH_ORDER, P4_WITH_ORDER | RELAX_TILES_64, P4_WITH_ORDER | RELAX_TILES_128, P4_WITH_ORDER | INDEX_ROWS):
        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu first fit + order in one workgroup, flags {flags:#x}")
        assert last_p4_order() == 1
    assert_same_outputs(fb, _want16(fb), emu_solve16(fb, flags=P4_WITH_ORDER | RELAX_TILES_64), "emu first fit + order in one workgroup, 16-bit cells")
    assert last_p4_order() == 1
    # where it does not apply the two kernels run as before: the ticket form asked for, the sampled verification, the general fill
    for flags in (P4_WITH_ORDER | TICKET_ORDER, P4_WITH_ORDER | RELAX_TILES_64 | (8 << 24), P4_WITH_ORDER | 1):
        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu, flags {flags:#x}")
        assert last_p4_order() == 0


def test_emu_first_fit_with_order_topics_failing_at_different_places():
    from emu_lib import P4_WITH_ORDER, last_p4_order
    fb = _multi_topic_scenarios(77, 3, 3, 700, 40, 8, 3)
    want = oracle_solve(fb)
    st = want.topic_results["status"]
    assert (st == abi.KAS_FAIL_UNASSIGNABLE).any() and (st == abi.KAS_SKIPPED).any() and (st == abi.KAS_OK).sum() >= 4
    for flags in (P4_WITH_ORDER | RELAX_TILES_64, P4_WITH_ORDER | RELAX_TILES_128):
        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), "emu first fit + order in one workgroup, several topics")
        assert last_p4_order() == 1


def test_emu_wide_lists_more_brokers_than_the_side_table_has_room_for():
    """Beyond ~6,500 brokers the wide ticket form's LDS has no room for the joint solve's front[] words
    (one per node): the kernel then runs without side dependencies — same results."""
    from kas_plan_math_py import wide_has_front
    fb = _batch(41, 1, 2500, 6560, 40, 5, ("add_k",))
    assert not wide_has_front(int(fb.scen["n_nodes"].max()))
    want = oracle_solve(fb)
    assert want.scenario_results["status"][0] == abi.KAS_OK
    assert_same_outputs(fb, want, emu_solve(fb), "emu wide tickets, no front[] table")


@pytest.mark.timeout(900)      # a field overlap shows as rows that never become ready
def test_emu_wide_lists_counts_near_the_field_limit():
    """Five-wide lists on few brokers: ~1000 rows per broker, so the 10-bit count fields of the wide
    ticket form — including the last one, next to the commits field — run up to their limit."""
    fb = _batch(31, 1, 3800, 20, 10, 5, ("add_k",), rack_aware=False)
    want = oracle_solve(fb)
    assert want.scenario_results["status"][0] == abi.KAS_OK
    out = want.out[:3800 * 5].reshape(3800, 5)
    assert np.bincount(out[:, 4]).max() > 255 and np.bincount(out.reshape(-1)).max() > 800
    assert_same_outputs(fb, want, emu_solve(fb), "emu wide, counts near the field limit")


@pytest.mark.timeout(900)
def test_emu_wide_lists_a_broker_holding_1023_rows_or_more_keeps_the_wide_ticket_form():
    """Round 3: up to 2,039 rows per broker the wide form runs and checks its 10-bit count fields when the last row
    has retired (they add up to the commits, field 2 stays below 1024).  Five-wide lists over 22 brokers, 4,600
    partitions: a broker holds 1,046 rows, at most 360 of them at one list position — nothing outgrows its field,
    the scenario is solved by the wide form alone."""
    from emu_lib import last_flagged, last_order_form, plan_shape
    fb = _batch(31, 1, 4600, 20, 10, 5, ("add_k",), rack_aware=False)
    rc, sh, err = plan_shape(fb)
    assert rc == 0 and sh["wide_ok"] == 1 and sh["wide_checked"] == 1, (rc, sh, err)
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_OK).all()
    assert np.bincount(want.out[:4600 * 5]).max() >= 1023
    got = emu_solve(fb)
    assert last_order_form() == 2 and last_flagged() == 0
    assert_same_outputs(fb, want, got, "emu wide, 1046 rows per broker")


def _wide_batch_whose_counts_outgrow_the_fields():
    """Scenario 0: one replica per partition in four-wide rows over two brokers (count[broker][0] reaches 1,050);
    scenario 1: a balanced two-replica assignment over nine brokers that nothing has to be moved in."""
    from kafka_assigner_amd.flatten import FlatBatch
    P, W = 2100, 4
    rng = np.random.default_rng(5)
    scen = np.zeros(2, dtype=abi.SCENARIO_DESC_DTYPE)
    topics = np.zeros(2, dtype=abi.TOPIC_DESC_DTYPE)
    node_id = np.array([7, 9, 1, 2, 3, 4, 5, 6, 8, 10, 12], dtype=np.int32)
    node_rack = np.array([0, 1, 0, 1, 2, 3, 4, 5, 6, 7, 8], dtype=np.int32)
    scen[0] = (2, 0, 1, 0, 0, -1)                  # two brokers, rf 1: 1,050 rows each, all at position 0
    scen[1] = (9, 1, 1, 0, 2, -1)                  # nine brokers, rf 2: 467 rows each
    topics[0] = (3644, P, 1, 1, W, 0, 0, 0, -1, -1, -1)                # (cur one wide, rf 1, rows of the batch's width)
    topics[1] = (3644, P, 2, 2, W, 0, P, P * W, -1, -1, -1)
    cur0 = rng.choice(np.array([7, 9, 11]), size=(P, 1)).astype(np.int32)   # (11 is gone: its rows are orphans)
    ids1 = np.array([1, 2, 3, 4, 5, 6, 8, 10, 12])
    rows = np.arange(P)
    cur1 = np.stack([ids1[rows % 9], ids1[(rows + 1 + (rows // 9) % 7) % 9]], axis=1).astype(np.int32)   # balanced: nothing moves
    fb = FlatBatch(scen=scen, topics=topics, node_id=node_id, node_rack=node_rack,
                   cur=np.concatenate([cur0.reshape(-1), cur1.reshape(-1)]).astype(np.int32),
                   aux=np.zeros(0, np.int32), ctx=np.zeros(0, np.int32), out_len=2 * P * W)
    return fb, P, W


@pytest.mark.timeout(900)
def test_emu_wide_lists_counts_that_outgrow_the_fields_are_found_and_the_scenario_is_solved_again():
    """One replica per partition in four-wide rows over two brokers: every row a broker holds counts at list position
    0, so count[broker][0] passes 1023 and carries into the next field.  The wide form finishes the scenario on wrong
    counts, its check at the end flags it, and the fill kernel and the round form solve it again from `cur` —
    beside a scenario of the same batch that stays inside the fields and is not touched."""
    from emu_lib import last_flagged, last_order_form, plan_shape
    fb, P, W = _wide_batch_whose_counts_outgrow_the_fields()
    rc, sh, err = plan_shape(fb)
    assert rc == 0 and sh["wide_ok"] == 1 and sh["wide_checked"] == 1, (rc, sh, err)
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_OK).all()
    assert np.bincount(want.out[:P * W].reshape(P, W)[:, 0]).max() >= 1024
    got = emu_solve(fb)
    assert last_order_form() == 2 and last_flagged() == 1
    assert_same_outputs(fb, want, got, "emu wide, count[.][0] beyond 1023")


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("N,P", [(5000, 12000), (7400, 9000)])
def test_emu_lists_3_wide_keep_the_ticket_form_up_to_8191_brokers(N, P):
    """Round 3: one scenario per solver wavefront is limited by where its counter rows end (8 B per broker below
    64 KiB), not by its whole LDS region: the ticket form serves up to 8,191 brokers (round 2: 4,680, the round
    form beyond — which does not even fit from 6,800 on)."""
    from emu_lib import last_order_form, plan_shape
    fb = _batch(4242, 2, P, N, 25, 3, ("add_k", "mixed"))
    rc, sh, err = plan_shape(fb)
    assert rc == 0 and sh["tickets_ok"] == 1 and sh["G"] == 1, (rc, sh, err)
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_OK).all()
    got = emu_solve(fb, flags=TICKET_ORDER)
    assert last_order_form() == 1
    assert_same_outputs(fb, want, got, f"emu ticket form, {N} brokers")
    got = emu_solve(fb)
    assert last_order_form() == 3
    assert_same_outputs(fb, want, got, f"emu relaxation form, {N} brokers")


SPREAD = 32        # KAS_PLAN_SPREAD_FILL


@pytest.mark.parametrize("S,P,N,R,RF,actions", [
    (3, 4000, 90, 18, 3, ("add_k", "remove1", "mixed")),
    (2, 1500, 100, 20, 5, ("add_k", "mixed")),
    (2, 1601, 80, 16, 4, G.ACTIONS),             # a last tile that is not full; stranding scenarios
])
def test_emu_spread_fill_matches_the_one_workgroup_fill(S, P, N, R, RF, actions):
    """The spread fill (row scans of a scenario over several one-wavefront workgroups, quota and P4 as
    kernels of their own) against the oracle and against the one-workgroup kernel, forced onto small
    single-topic batches."""
    from emu_lib import last_spread
    fb = _batch(200 + RF, S, P, N, R, RF, actions)
    want = oracle_solve(fb)
    assert_same_outputs(fb, want, emu_solve(fb, flags=SPREAD), "emu spread fill")
    assert last_spread() == S
    assert_same_outputs(fb, want, emu_solve(fb), "emu one-workgroup fill")
    assert last_spread() == 0


def test_emu_spread_fill_hands_back_what_it_does_not_cover():
    """Rows that are not rack-diverse (the general sticky fill's case), rack awareness off, and a batch
    with a multi-topic scenario: the spread fill hands such scenarios (or the whole batch) back."""
    from emu_lib import last_spread
    fb = _batch(77, 3, 2000, 40, 2, 3, ("remove1", "add_k"), cyclic=True)      # cyclic start on 2 racks: co-racked replicas
    want = oracle_solve(fb)
    assert_same_outputs(fb, want, emu_solve(fb, flags=SPREAD), "emu spread fill, rows not rack-diverse")
    assert last_spread() == 0
    fb = _batch(78, 2, 3000, 60, 12, 3, ("add_k",), rack_aware=False)
    assert_same_outputs(fb, oracle_solve(fb), emu_solve(fb, flags=SPREAD), "emu spread fill, rack awareness off")
    assert last_spread() == 2
    fb = _multi_topic_scenarios(3, 2, 3, 900, 40, 8, 3)
    assert_same_outputs(fb, oracle_solve(fb), emu_solve(fb, flags=SPREAD), "emu spread flag, multi-topic batch")
    assert last_spread() == 0


# ---- ticket forms with a Context handed in (KAS:59-62; how KTA:70-71 calls the solver on EVERY topic) ----
def _with_context(fb, values):
    """Give every scenario of a single-topic batch a Context: ctx_width 8, counters from values(s, n_nodes)."""
    from kafka_assigner_amd.flatten import FlatBatch
    scen = fb.scen.copy()
    ctx, off = [], 0
    for s in range(fb.n_scenarios):
        n = int(scen["n_nodes"][s])
        tab = np.asarray(values(s, n), dtype=np.int32).reshape(n, 8)
        scen["ctx_width"][s] = 8
        scen["ctx_off"][s] = off
        ctx.append(tab.reshape(-1)); off += n * 8
    return FlatBatch(scen=scen, topics=fb.topics, node_id=fb.node_id, node_rack=fb.node_rack, cur=fb.cur, aux=fb.aux,
                     ctx=np.concatenate(ctx), out_len=fb.out_len)


def _random_counters(seed, hi):
    def values(s, n):
        return np.random.default_rng(seed + s).integers(0, hi, size=(n, 8))
    return values


@pytest.mark.parametrize("P,N,R,RF,form", [(3000, 60, 10, 3, 1), (1500, 40, 8, 2, 1), (1200, 50, 10, 5, 2), (900, 40, 8, 4, 2)])
def test_emu_ticket_forms_take_a_context_in_and_hand_it_back(P, N, R, RF, form):
    """A Context no longer sends a scenario to the one-wavefront round form: the ticket kernels seed
    their count fields from it and write them back (the counters of columns >= list width pass through
    untouched).  Same lists, same counters as the oracle — and the plan really chose the ticket form."""
    from emu_lib import last_flagged, last_order_form
    fb = _with_context(_batch(808 + RF, 4, P, N, R, RF, G.BENCH_ACTIONS), _random_counters(5, 60))
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_OK).any()
    got = emu_solve(fb, flags=TICKET_ORDER)
    assert last_order_form() == form and last_flagged() == 0
    assert_same_outputs(fb, want, got, f"emu ticket form {form} with a Context")
    assert (got.ctx != fb.ctx).any()                                     # the counters moved ...
    np.testing.assert_array_equal(got.ctx.reshape(-1, 8)[:, RF:], fb.ctx.reshape(-1, 8)[:, RF:])   # ... only below the list width
    assert_same_outputs(fb, want, emu_solve(fb, flags=2), "emu round form with the same Context")
    if RF <= 3:
        # round 4: lists up to 3 wide take the relaxation form with a Context too (counter words seeded from its
        # columns 0 and 1, column 2 counted beside them), at both tile sizes
        for flags, what in ((0, "by batch size"), (RELAX_TILES_64, "tiles of 64 rows"), (RELAX_TILES_128, "double tiles")):
            got = emu_solve(fb, flags=flags)
            assert last_order_form() == 3 and last_flagged() == 0
            assert_same_outputs(fb, want, got, f"emu relaxation form with a Context, {what}")
            np.testing.assert_array_equal(got.ctx.reshape(-1, 8)[:, RF:], fb.ctx.reshape(-1, 8)[:, RF:])


def test_emu_context_counters_beyond_the_count_fields_go_to_the_round_form():
    """Per scenario, on the device: largest counter + rows a node can gain must stay inside the count
    fields (16 bits for lists <= 3 wide, 10 bits for the wide form), else the scenario is flagged and the
    round form (int32 counters), launched behind the ticket kernel, solves it.  Two of four scenarios
    each; a negative counter (never produced by the reference, but an int in its map) counts as large."""
    from emu_lib import last_flagged, last_order_form

    def big3(s, n):
        v = np.random.default_rng(s).integers(0, 500, size=(n, 8))
        if s == 1: v[n // 2, 1] = 65535
        if s == 2: v[3, 0] = -4
        return v
    fb = _with_context(_batch(515, 4, 2000, 50, 10, 3, G.BENCH_ACTIONS), big3)
    want = oracle_solve(fb)
    got = emu_solve(fb, flags=1 << 12)
    assert last_order_form() == 1 and last_flagged() == 2
    assert_same_outputs(fb, want, got, "emu ticket form, two scenarios flagged")
    assert_same_outputs(fb, want, emu_solve(fb, flags=TICKET_ORDER), "emu ticket form, two scenarios per wavefront, two flagged")
    # the relaxation form keeps columns 0 and 1 in 16-bit fields: counter + rows to come must stay below 65536
    def big12(s, n):
        v = np.random.default_rng(s).integers(0, 500, size=(n, 8))
        if s == 0: v[n // 3, 0] = 65500                                 # + up to 150 rows: over
        if s == 1: v[n // 2, 1] = 65535
        if s == 2: v[3, 0] = -4
        if s == 3: v[5, 2] = 1 << 30                                    # column 2 is not a field: stays
        return v
    # (1 << 30 in a counter the round form compares: its pick keys are 64 bits wide — round 4's random stress found the
    # 32-bit `count << 3` of earlier rounds wrapping from 2^28 on)
    def huge(s, n):
        v = np.random.default_rng(s).integers(0, 50, size=(n, 8))
        v[n // 2, s % 3] = (1 << 29) + s
        if s == 1: v[2, 1] = (1 << 30) + 5
        return v
    fbh = _with_context(_batch(517, 4, 900, 40, 8, 3, G.BENCH_ACTIONS), huge)
    wanth = oracle_solve(fbh)
    for flags in (0, TICKET_ORDER, 2):
        assert_same_outputs(fbh, wanth, emu_solve(fbh, flags=flags), f"emu Context counters of 2^29 and 2^30, flags {flags:#x}")
    fb12 = _with_context(_batch(515, 4, 2000, 50, 10, 3, G.BENCH_ACTIONS), big12)
    want12 = oracle_solve(fb12)
    got = emu_solve(fb12)
    assert last_order_form() == 3 and last_flagged() == 3
    assert_same_outputs(fb12, want12, got, "emu relaxation form, three scenarios flagged")

    def big5(s, n):
        v = np.random.default_rng(s).integers(0, 300, size=(n, 8))
        if s == 0: v[1, 4] = 1000
        if s == 3: v[n - 1, 2] = 70000
        return v
    fb = _with_context(_batch(516, 4, 1000, 50, 10, 5, ("add_k", "mixed")), big5)
    want = oracle_solve(fb)
    got = emu_solve(fb)
    assert last_order_form() == 2 and last_flagged() == 2
    assert_same_outputs(fb, want, got, "emu wide ticket form, two scenarios flagged")


def test_emu_per_topic_calls_carry_the_context_like_the_cli_loop():
    """KAG:172-184 through the per-topic drop-in: one call per topic, the Context of call k is the
    input of call k + 1 — every call takes the ticket form, and the chain equals one oracle run over
    the whole scenario."""
    from emu_lib import last_order_form
    N, R, RF = 40, 8, 3
    act, bs = G.scenario_action(9, 0, N, R, actions=("remove1",), max_add=4)
    racks = {int(b): "r%d" % int(r) for b, r in zip(bs.node_id, bs.node_rack)}
    brokers = [int(b) for b in bs.node_id]
    topics = [Topic("topic-%d" % t, {p: row.tolist() for p, row in enumerate(G.random_assignment(40 + t, 500 + 37 * t, N, R, RF))}, RF)
              for t in range(4)]
    whole = flatten([Scenario(brokers=brokers, racks=racks, topics=topics, want_context=True)])
    want = oracle_solve(whole)
    assert (want.topic_results["status"][:2] == abi.KAS_OK).all()
    from kafka_assigner_amd.flatten import unflatten_context
    for flags, form in ((0, 3), (TICKET_ORDER, 1)):                      # (round 4: the relaxation form; round 3: tickets)
        counters, outs = None, []
        for k, t in enumerate(topics):
            fb = flatten([Scenario(brokers=brokers, racks=racks, topics=[t], context=counters, want_context=True)])
            got = emu_solve(fb, flags=flags)
            assert last_order_form() == form
            assert got.topic_results["status"][0] == want.topic_results["status"][k]
            assert got.topic_results["fail_partition"][0] == want.topic_results["fail_partition"][k]
            outs.append(got.out[:fb.out_len])
            if got.topic_results["status"][0] != abi.KAS_OK:
                break                                                    # the CLI run ends here (KAG:173-184)
            counters = unflatten_context(fb, got.ctx, 0)
        done = np.concatenate(outs)
        np.testing.assert_array_equal(done, want.out[:done.size])
        assert counters == unflatten_context(whole, want.ctx, 0)


def test_emu_many_rows_per_broker_keep_the_relaxation_form_up_to_16_bit_counts():
    """Round 4, last session: the relaxation form's count fields are 16 bits wide (the first version had 12 and handed
    a broker with 4,095 rows or more to the ticket form).  90,000 partitions on 63 brokers: 4,286 rows per broker;
    70,000 partitions on 3 brokers at RF 3: every broker holds every row — beyond the fields, the round form."""
    from emu_lib import last_order_form, plan_shape
    bs = G.perturb_brokers(60, 10, add=3, rack_aware=False)              # (with racks the reference strands at this density)
    fb = uniform_batch(G.random_assignment(61, 90000, 60, 10, 3)[None], bs.node_id[None], bs.node_rack[None], 3)
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_OK).all()
    assert np.bincount(want.out[:90000 * 3]).max() > 4095
    for flags in (0, RELAX_TILES_64, TICKET_ORDER):
        got = emu_solve(fb, flags=flags)
        assert last_order_form() == (1 if flags == TICKET_ORDER else 3)
        assert_same_outputs(fb, want, got, f"emu 4,286 rows per broker, flags {flags:#x}")
    bs3 = G.perturb_brokers(3, 3)
    fb = uniform_batch(G.random_assignment(62, 70000, 3, 3, 3)[None], bs3.node_id[None], bs3.node_rack[None], 3)
    rc, sh, _ = plan_shape(fb)
    assert rc == 0 and sh["relax_ok"] == 0 and sh["tickets_ok"] == 0
    assert_same_outputs(fb, oracle_solve(fb), emu_solve(fb), "emu 70k rows per broker: round form")
    assert last_order_form() == 0


def test_emu_mixed_batch_over_tiles_of_64_rows():
    """One batch, every plan variant: a scenario whose rows are not rack-diverse (the general fill) beside rack-diverse ones,
    scenarios with several topics of which one fails (the topics behind it are skipped), a topic 2 wide beside topics 3
    wide, rf raised so that EVERY row is an orphan, and ragged last tiles.  (Written for round 5's orphan-record experiment,
    experiments/README.md; kept because nothing else mixes these in one launch.)"""
    rng = np.random.default_rng(5)
    scs = []
    N, R = 48, 8
    racks = {b: "r%d" % (b % R) for b in range(N + 4)}
    def topic(name, seed, P, rf, cyc=False):
        cur = G.cyclic_assignment(P, N, rf, seed) if cyc else G.random_assignment(seed, P, N, R, rf)
        return Topic(name, {p: cur[p].tolist() for p in range(P)}, rf)
    scs.append(Scenario(brokers=[b for b in range(N) if b != 7], racks=racks,
                        topics=[topic("t-a", 1, 1500, 3), topic("t-b", 2, 333, 2), topic("t-c", 3, 700, 3)]))
    scs.append(Scenario(brokers=[b for b in range(N) if b != 13], racks=racks, topics=[topic("u-a", 7, 2100, 3), topic("u-b", 5, 90, 3)]))
    # rows made for racks b mod 8 on a cluster whose racks are b mod 7: co-racked replicas -> the general fill, inside a launch
    # that writes records for the others
    scs.append(Scenario(brokers=list(range(N)), racks={b: "q%d" % (b % 7) for b in range(N)}, topics=[topic("v-a", 6, 800, 3)]))
    # rf 2 -> 3 at lists 3 wide: every row needs one more replica (KTA:57, Q6)
    cur2 = G.random_assignment(9, 900, N, R, 2)
    scs.append(Scenario(brokers=list(range(N)), racks=racks, topics=[Topic("w-a", {p: cur2[p].tolist() for p in range(900)}, 3)]))
    # a decommission that cannot be absorbed: the first topic fails in P4, the next one is skipped
    few = list(range(6))
    cur3 = G.cyclic_assignment(400, 6, 3, 0)
    scs.append(Scenario(brokers=few[:5], racks={b: "abc"[b % 3] for b in few},
                        topics=[Topic("x-a", {p: cur3[p].tolist() for p in range(400)}, 3), topic("x-b", 11, 200, 3)]))
    fb = flatten(scs)
    want = oracle_solve(fb)
    st = want.topic_results["status"]
    assert st.tolist() == [0, 0, 0, 0, 0, 0, 0, 1, 6], st.tolist()
    assert want.topic_results["moved_replicas"][6] == 900      # (w-a: every row an orphan)
    assert want.topic_results["moved_replicas"][5] > 0         # (v-a: the general fill moved something)
    for flags, split in ((RELAX_TILES_64, 1), (RELAX_TILES_64 | NO_RTN_QUOTA, 1), (RELAX_TILES_64 | (2 << 8), 0), (1 << 8, 0), (8, 1), (0, 1),
                         (FILL_WITH_P4, 0), (FILL_WITH_P4 | RELAX_TILES_64, 0), (TICKET_ORDER, 1), (2, 1), (1, 0)):
        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), "emu, flags %#x" % flags)
        assert last_split_p4() == split, hex(flags)         # (first fit in kas_p4_kernel: four fill wavefronts, rack-diverse form)


def test_first_fit_takes_its_own_kernel_from_512_scenarios_on():
    """kas_split_p4: by batch size unless the plan flags name a form."""
    for S, split in ((511, 0), (512, 1)):
        fb = _batch(808, S, 130, 12, 4, 3, ("remove1", "add_k"))
        want = oracle_solve(fb, threads=0)
        assert_same_outputs(fb, want, emu_solve(fb, p4_by_batch_size=True), "emu, %d scenarios" % S)
        assert last_split_p4() == split, S
    assert_same_outputs(fb, want, emu_solve(fb, flags=FILL_WITH_P4), "emu, 512 scenarios, first fit inside the fill workgroup")
    assert last_split_p4() == 0


def test_emu_dword_mid_rows_where_they_apply_and_where_they_do_not():
    """KAS_FLAG_MID32 (round 6): between the fill and the order kernel a row of up to three holders is ONE dword — the holders
    sorted by node index, 11 bits each (first fit appends, KAS:228 sorts: nothing reads their order) — where the 16-bit rows are
    three uint16 in acceptance order.  int32 cells, lists 3 wide, at most 2,047 brokers, the relaxation form without a Context or
    the sampled verification; every place first fit can run in, the scenarios the slim kernel hands back (general fill, first fit
    from the ring), topics narrower than the batch, failures; both layouts against the oracle."""
    from emu_lib import FULL_FILL, MID32, NO_MID32, P4_WITH_ORDER, SPLIT_P4, last_mid32, last_slim_fill
    fb = _batch(4321, 5, 1777, 45, 9, 3, G.BENCH_ACTIONS)
    want = oracle_solve(fb)
    for flags, m in ((0, 1), (MID32, 1), (NO_MID32, 0), (P4_WITH_ORDER, 1), (P4_WITH_ORDER | RELAX_TILES_128, 1), (SPLIT_P4 | RELAX_TILES_64, 1),
                     (FILL_WITH_P4, 1), (FILL_WITH_P4 | RELAX_TILES_128, 1), (FULL_FILL, 1), (FULL_FILL | SPLIT_P4, 1), (NO_RTN_QUOTA, 1), (8, 1), (1, 1),
                     (2 << 8, 1), (1 << 8, 1), (INDEX_ROWS, 0), (TICKET_ORDER, 0), (2, 0), (7 << 24, 0)):
        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu dword mid rows, plan flags {flags:#x}")
        assert last_mid32() == m, (hex(flags), last_mid32())
    # a Context handed in or wanted back: the 16-bit rows
    fbc = flatten([Scenario(brokers=list(range(30)), racks={b: "r%d" % (b % 6) for b in range(30)}, want_context=True,
                            topics=[Topic("a", {p: G.random_assignment(1, 700, 30, 6, 3)[p].tolist() for p in range(700)}, 3)])])
    assert_same_outputs(fbc, oracle_solve(fbc), emu_solve(fbc), "emu with a Context")
    assert last_mid32() == 0
    # lists 2 wide: a row is one dword already
    fb2 = _batch(77, 3, 900, 30, 6, 2, G.ACTIONS)
    assert_same_outputs(fb2, oracle_solve(fb2), emu_solve(fb2), "emu lists 2 wide")
    assert last_mid32() == 0
    # rows that are not rack-diverse (scenarios 1 and 3: handed back, general fill, first fit from the ring), brokers that left
    racks = (np.arange(60) % 6).astype(np.int32)
    ids = np.arange(60, dtype=np.int32)
    curs = [G.random_assignment(5, 1200, 60, 6, 3), G.cyclic_assignment(1200, 60, 3, 1) * 6 % 60,
            G.random_assignment(6, 1200, 60, 6, 3), G.cyclic_assignment(1200, 60, 3, 2) * 6 % 60]
    fbm = uniform_batch(np.stack(curs).astype(np.int32), np.tile(ids, (4, 1))[:, :58], np.tile(racks, (4, 1))[:, :58], 3)
    wantm = oracle_solve(fbm)
    for flags in (0, P4_WITH_ORDER, FILL_WITH_P4, FULL_FILL, 1, NO_MID32):
        assert_same_outputs(fbm, wantm, emu_solve(fbm, flags=flags), f"emu dword mid rows: two of four scenarios not rack-diverse, plan flags {flags:#x}")
        assert last_mid32() == (0 if flags == NO_MID32 else 1)
    # duplicates in a row, brokers that are not in the set, ids far from zero, ragged last tile
    rng = np.random.default_rng(61)
    P, N = 1333, 37
    ids2 = (np.arange(N, dtype=np.int32) * 3 + 1000)[None, :]
    racks2 = (np.arange(N) % 7).astype(np.int32)[None, :]
    cur = (rng.integers(0, N + 6, size=(P, 3)).astype(np.int32) * 3 + 1000)
    cur[::17, 1] = cur[::17, 0]
    fbd = uniform_batch(cur[None], ids2, racks2, 3)
    assert_same_outputs(fbd, oracle_solve(fbd), emu_solve(fbd), "emu dword mid rows: duplicates, absent brokers")
    assert last_mid32() == 1
    # topics of several widths in one scenario (narrower topics are one dword a row too), a partition subset, a reduced RF
    scs = [Scenario(brokers=list(range(30)), racks={b: "r%d" % (b % 6) for b in range(30)}, want_context=False,
                    topics=[Topic("a", {p: G.random_assignment(1, 700, 30, 6, 3)[p].tolist() for p in range(700)}, 3),
                            Topic("b", {p: G.random_assignment(2, 500, 30, 6, 2)[p].tolist() for p in range(500)}, 3),
                            Topic("c", {p: G.random_assignment(3, 300, 30, 6, 2)[p].tolist() for p in range(300)}, 2),
                            Topic("e", {p: G.random_assignment(7, 130, 30, 6, 1)[p].tolist() for p in range(130)}, 1),
                            Topic("f", {p: G.random_assignment(8, 400, 30, 6, 3)[p].tolist() for p in range(400)}, 2),
                            Topic("d", {p: G.random_assignment(4, 900, 30, 6, 3)[p].tolist() for p in range(900)}, 3)])
           for _ in range(2)]
    scs[1].brokers = [b for b in range(30) if b not in (3, 17)]
    scs[1].racks = {b: "r%d" % (b % 6) for b in scs[1].brokers}
    fbw = flatten(scs)
    wantw = oracle_solve(fbw)
    for flags in (0, P4_WITH_ORDER, FILL_WITH_P4, SPLIT_P4 | RELAX_TILES_64, NO_MID32):
        assert_same_outputs(fbw, wantw, emu_solve(fbw, flags=flags), f"emu dword mid rows, topics of several widths, plan flags {flags:#x}")
    # first fit fails a topic (KAS:183-184: replace one broker at zero slack), in every place it runs in
    fbf = _batch(99, 6, 8000, 80, 8, 3, ("replace1",))
    wantf = oracle_solve(fbf)
    assert (wantf.scenario_results["status"] != abi.KAS_OK).any()
    for flags in (0, P4_WITH_ORDER, FILL_WITH_P4, SPLIT_P4):
        assert_same_outputs(fbf, wantf, emu_solve(fbf, flags=flags), f"emu dword mid rows, failing scenarios, plan flags {flags:#x}")
        assert last_mid32() == 1


def dword_mid_row_cases():
    """(what, batch, plan flag words) — batches in which the dword mid rows meet everything that touches a mid row; shared with the
    GPU test (tests/test_hip_parity.py)."""
    from emu_lib import FULL_FILL, NO_MID32, P4_WITH_ORDER, RELAX_TILES_256, SPLIT_P4
    yield ("bench mix", _batch(4321, 5, 1777, 45, 9, 3, G.BENCH_ACTIONS),
           (0, NO_MID32, P4_WITH_ORDER, RELAX_TILES_256, P4_WITH_ORDER | RELAX_TILES_256, FILL_WITH_P4 | RELAX_TILES_256, NO_MID32 | RELAX_TILES_256, P4_WITH_ORDER | RELAX_TILES_128, SPLIT_P4 | RELAX_TILES_64, FILL_WITH_P4, FILL_WITH_P4 | RELAX_TILES_128, FULL_FILL,
            FULL_FILL | SPLIT_P4, NO_RTN_QUOTA, 8, 1, 2 << 8, 1 << 8))
    racks = (np.arange(60) % 6).astype(np.int32)
    ids = np.arange(60, dtype=np.int32)
    curs = [G.random_assignment(5, 1200, 60, 6, 3), G.cyclic_assignment(1200, 60, 3, 1) * 6 % 60,
            G.random_assignment(6, 1200, 60, 6, 3), G.cyclic_assignment(1200, 60, 3, 2) * 6 % 60]
    yield ("two of four scenarios not rack-diverse", uniform_batch(np.stack(curs).astype(np.int32), np.tile(ids, (4, 1))[:, :58], np.tile(racks, (4, 1))[:, :58], 3),
           (0, P4_WITH_ORDER, FILL_WITH_P4, FULL_FILL, 1, NO_MID32))
    rng = np.random.default_rng(61)
    P, N = 1333, 37
    cur = (rng.integers(0, N + 6, size=(P, 3)).astype(np.int32) * 3 + 1000)
    cur[::17, 1] = cur[::17, 0]
    yield ("duplicates, absent brokers", uniform_batch(cur[None], (np.arange(N, dtype=np.int32) * 3 + 1000)[None, :], (np.arange(N) % 7).astype(np.int32)[None, :], 3),
           (0, P4_WITH_ORDER, SPLIT_P4))
    scs = [Scenario(brokers=list(range(30)), racks={b: "r%d" % (b % 6) for b in range(30)}, want_context=False,
                    topics=[Topic("a", {p: G.random_assignment(1, 700, 30, 6, 3)[p].tolist() for p in range(700)}, 3),
                            Topic("b", {p: G.random_assignment(2, 500, 30, 6, 2)[p].tolist() for p in range(500)}, 3),
                            Topic("c", {p: G.random_assignment(3, 300, 30, 6, 2)[p].tolist() for p in range(300)}, 2),
                            Topic("e", {p: G.random_assignment(7, 130, 30, 6, 1)[p].tolist() for p in range(130)}, 1),
                            Topic("f", {p: G.random_assignment(8, 400, 30, 6, 3)[p].tolist() for p in range(400)}, 2),
                            Topic("d", {p: G.random_assignment(4, 900, 30, 6, 3)[p].tolist() for p in range(900)}, 3)])
           for _ in range(2)]
    scs[1].brokers = [b for b in range(30) if b not in (3, 17)]
    scs[1].racks = {b: "r%d" % (b % 6) for b in scs[1].brokers}
    yield ("topics of several widths", flatten(scs), (0, P4_WITH_ORDER, FILL_WITH_P4, SPLIT_P4 | RELAX_TILES_64, NO_MID32, P4_WITH_ORDER | RELAX_TILES_256))
    yield ("failing scenarios (replace one broker at zero slack)", _batch(99, 6, 8000, 80, 8, 3, ("replace1",)), (0, P4_WITH_ORDER, FILL_WITH_P4, SPLIT_P4, P4_WITH_ORDER | RELAX_TILES_256, RELAX_TILES_256))
    for N, R, P in ((1023, 11, 4000), (1025, 25, 4000), (2047, 23, 5000), (2048, 32, 5000)):
        cur = G.random_assignment(N, P, N, R, 3)
        ids = np.arange(N + 40, dtype=np.int32)
        keep = np.ones(N + 40, dtype=bool); keep[[5, 1000 % N, N - 2]] = False; keep[N:] = False
        keep[N:N + 7] = True                                                         # seven brokers join, three leave
        bs = ids[keep]
        if N == 2047:
            bs = bs[:2047]
        yield (f"{bs.shape[0]} brokers", uniform_batch(cur.astype(np.int32)[None], bs[None, :], (bs % R).astype(np.int32)[None, :], 3), (0, P4_WITH_ORDER, NO_MID32, P4_WITH_ORDER | RELAX_TILES_256))


def test_emu_dword_mid_rows_at_the_field_limits():
    """Node indices around 1,024 (where the packed fields change roles) and up to 2,046; 2,048 brokers: the 16-bit rows."""
    from emu_lib import NO_MID32, P4_WITH_ORDER, last_mid32
    for N, R, P, want_m in ((1023, 11, 4000, 1), (1025, 25, 4000, 1), (2047, 23, 5000, 1), (2048, 32, 5000, 0)):
        cur = G.random_assignment(N, P, N, R, 3)
        # (rows across the 1,024 boundary in every combination: low / mid / high indices)
        cur[::3, 0] = np.minimum(cur[::3, 0], N - 1)
        ids = np.arange(N + 40, dtype=np.int32)
        keep = np.ones(N + 40, dtype=bool); keep[[5, 1000 % N, N - 2]] = False; keep[N:] = False
        keep[N:N + 7] = True                                                         # seven brokers join, three leave
        bs = ids[keep]
        racks = (bs % R).astype(np.int32)
        if bs.shape[0] > 2047 and want_m:
            bs, racks = bs[:2047], racks[:2047]
        fb = uniform_batch(cur.astype(np.int32)[None], bs[None, :], racks[None, :], 3)
        want = oracle_solve(fb)
        for flags in (0, P4_WITH_ORDER):
            assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu dword mid rows, {bs.shape[0]} brokers, plan flags {flags:#x}")
            assert last_mid32() == (1 if bs.shape[0] <= 2047 else 0), (N, bs.shape[0], last_mid32())
        assert_same_outputs(fb, want, emu_solve(fb, flags=NO_MID32), f"emu 16-bit mid rows, {bs.shape[0]} brokers")


def test_emu_dword_mid_row_cases_shared_with_the_gpu_test():
    """... among the plan flag words: KAS_PLAN_RELAX_TILES(3), quad tiles (256 rows a step) — the instances exist on dword mid rows;
    anywhere else the launch keeps double tiles"""
    from emu_lib import NO_MID32, RELAX_TILES_256, last_mid32, last_relax_quad
    for what, fb, flag_words in dword_mid_row_cases():
        want = oracle_solve(fb)
        fits = int(fb.scen["n_nodes"].max()) <= 2047
        for flags in flag_words:
            assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu dword mid rows: {what}, plan flags {flags:#x}")
            assert last_mid32() == (1 if fits and not (flags & NO_MID32) else 0), (what, hex(flags))
            assert last_relax_quad() == (1 if (flags & RELAX_TILES_256) == RELAX_TILES_256 and last_mid32() else 0), (what, hex(flags))
