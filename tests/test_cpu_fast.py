"""oracle/kas_cpu_fast.c (CPU baseline B2: flat arrays, full-node skipping) must produce exactly
what the oracle produces — it is timed beside the GPU path as "what a host can do", so its
results have to be the reference's.  Also: the threaded entries (scenario-parallel pthreads inside
one C call) of both CPU solvers return what the single-thread entries return."""
import numpy as np
from hypothesis import HealthCheck, given, seed, settings

from kafka_assigner_amd import abi
from kafka_assigner_amd import generator as G
from kafka_assigner_amd.flatten import Scenario, Topic, flatten, uniform_batch
from oracle_lib import cpu_fast_solve, host_threads, oracle_solve
from parity_util import assert_same_outputs
from test_emu_parity import _batch, _multi_topic_scenarios
from test_oracle_vs_literal import scenarios


@seed(20260923)
@settings(max_examples=400, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(scenarios())
def test_cpu_fast_equals_oracle_small_odd_inputs(sc):
    """Ragged lists, duplicate brokers, sparse ids, partitions != keys(cur), RF up/down, the
    Integer.MIN_VALUE hash, multi-topic Context carry-over, failures and skips."""
    brokers, racks, topics = sc
    for want_ctx in (True, False):
        fb = flatten([Scenario(brokers=brokers, racks=racks, want_context=want_ctx,
                               topics=[Topic(n, c, rf, parts) for n, c, rf, parts in topics])])
        assert_same_outputs(fb, oracle_solve(fb), cpu_fast_solve(fb), "cpu_fast")


def test_cpu_fast_equals_oracle_seeded_batches_every_action():
    for P, N, R, RF, actions in ((1000, 40, 8, 3, G.ACTIONS), (3000, 100, 10, 3, ("remove1",)),
                                 (777, 40, 10, 5, G.ACTIONS), (640, 24, 8, 4, ("replace1", "remove1")),
                                 (8000, 80, 8, 3, ("replace1", "add_k", "mixed"))):
        fb = _batch(1234, 6, P, N, R, RF, actions)
        want = oracle_solve(fb)
        assert_same_outputs(fb, want, cpu_fast_solve(fb), f"cpu_fast {P}x{N}")
    fb = _batch(99, 4, 1500, 50, 10, 3, G.ACTIONS, rack_aware=False)
    assert_same_outputs(fb, oracle_solve(fb), cpu_fast_solve(fb), "cpu_fast norack")
    fb = _batch(7, 4, 1200, 60, 6, 3, ("add_k", "remove1"), cyclic=True)      # strandings: same failing partition
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_FAIL_UNASSIGNABLE).any()
    assert_same_outputs(fb, want, cpu_fast_solve(fb), "cpu_fast cyclic")
    fb = _multi_topic_scenarios(77, 3, 3, 700, 40, 8, 3)
    assert_same_outputs(fb, oracle_solve(fb), cpu_fast_solve(fb), "cpu_fast multi-topic")


def test_cpu_fast_sparse_ids_fall_back_to_binary_search():
    cur = G.random_assignment(5, 500, 20, 5, 3).astype(np.int64) * 100003 + 7
    ids = (np.arange(20, dtype=np.int64) * 100003 + 7).astype(np.int32)[None, :]
    racks = (np.arange(20) % 5).astype(np.int32)[None, :]
    fb = uniform_batch(cur.astype(np.int32)[None], ids[:, :19], racks[:, :19], 3)
    assert_same_outputs(fb, oracle_solve(fb), cpu_fast_solve(fb), "cpu_fast sparse")


def test_cpu_fast_config3_and_config5_shapes():
    """One full-size C3 scenario per action kind and a C5-shaped RF 5 scenario (5k brokers)."""
    fb = _batch(2024, 4, 100000, 1000, 20, 3, G.ACTIONS)
    assert_same_outputs(fb, oracle_solve(fb), cpu_fast_solve(fb), "cpu_fast C3")
    P, N, R, RF = 60000, 5000, 40, 5
    cur = G.random_assignment(7, P, N, R, RF)
    for rack_aware in (True, False):
        bs = G.perturb_brokers(N, R, remove=list(range(0, N, 50)), add=200, rack_aware=rack_aware)
        fb = uniform_batch(cur[None], bs.node_id[None], bs.node_rack[None], RF)
        assert_same_outputs(fb, oracle_solve(fb), cpu_fast_solve(fb), f"cpu_fast C5 rack_aware={rack_aware}")


def test_threaded_entries_return_what_the_single_thread_entries_return():
    fb = _batch(31, 13, 2500, 60, 6, 3, G.ACTIONS)            # more scenarios than most thread counts divide
    want = oracle_solve(fb)
    assert host_threads() >= 1
    for threads in (2, 5, 0):
        got = oracle_solve(fb, threads=threads)
        assert 1 <= got.threads_used <= max(13, host_threads())
        assert_same_outputs(fb, want, got, f"oracle, {threads} threads")
        assert_same_outputs(fb, want, cpu_fast_solve(fb, threads=threads), f"cpu_fast, {threads} threads")
    # multi-topic scenarios with a Context in/out: a scenario stays on one thread
    scs = []
    for s in range(5):
        cur = G.random_assignment(40 + s, 300, 12, 4, 3)
        scs.append(Scenario(brokers=list(range(12)) + [20 + s], racks={b: "r%d" % (b % 4) for b in range(40)},
                            want_context=True,
                            topics=[Topic("t%d" % t, {p: cur[p].tolist() for p in range(300)}, 3) for t in range(3)]))
    fb = flatten(scs)
    assert_same_outputs(fb, oracle_solve(fb), oracle_solve(fb, threads=3), "oracle ctx, 3 threads")
    assert_same_outputs(fb, oracle_solve(fb), cpu_fast_solve(fb, threads=3), "cpu_fast ctx, 3 threads")
