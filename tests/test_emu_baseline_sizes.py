"""The kernel source at BASELINE.json's STATED sizes, on the CPU: csrc/kas_solver_body.h and kas_order_wide.h stepped
by the fiber emulator (tests/emu, the product's own plan math choosing the kernels) against the oracle, every list.
The GPU suite runs the same configurations through the C ABI (tests/test_hip_parity.py); these need no GPU, so the
`-m "not gpu"` run also covers configs[1]..[4] at full size since the emulator's fibers stopped costing a system
call per switch (an emulated 100k x 1k scenario takes ~0.5 s, the 1M x 5k x RF 5 one ~30 s)."""
import numpy as np
import pytest

from kafka_assigner_amd import abi
from kafka_assigner_amd import generator as G
from kafka_assigner_amd.flatten import uniform_batch
from emu_lib import RELAX_TILES_64, RELAX_TILES_128, TICKET_ORDER, emu_solve, last_order_form, last_queue_rows, last_relax_stats, last_spread
from oracle_lib import oracle_solve
from parity_util import assert_same_outputs
from test_emu_parity import _batch


def test_emu_config2_10k_partitions_100_brokers_decommission_one():
    """configs[1]: 10k partitions x 100 brokers x 10 racks, RF 3, remove broker s mod 100."""
    for seed in (0, 1, 2, 3):
        cur = G.random_assignment(seed, 10000, 100, 10, 3)
        bs = G.perturb_brokers(100, 10, remove=[seed % 100])
        fb = uniform_batch(cur[None], bs.node_id[None], bs.node_rack[None], 3)
        want = oracle_solve(fb)
        assert want.scenario_results["status"][0] == abi.KAS_OK
        assert_same_outputs(fb, want, emu_solve(fb), f"emu C2 seed {seed}")
        assert_same_outputs(fb, want, emu_solve(fb, flags=TICKET_ORDER), f"emu C2 seed {seed}, ticket form")
        assert_same_outputs(fb, want, emu_solve(fb, flags=2), f"emu C2 seed {seed}, round form")


def test_emu_config3_full_size_scenarios_every_action_every_plan_variant():
    """configs[2]'s scenario: 100k partitions x 1k brokers x 20 racks, RF 3, the four action kinds — the kernels the
    bench launches (fill with per-chunk histograms + the relaxation form of the order kernel) and every other form the
    plan can be made to take (the 3-wide ticket form, two scenarios per wavefront, packed counter rows, first)."""
    fb = _batch(2024, 4, 100000, 1000, 20, 3, G.ACTIONS)
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_OK).sum() >= 2
    assert_same_outputs(fb, want, emu_solve(fb), "emu C3")
    assert last_order_form() == 3, "not the relaxation form"
    tiles, evals, slow = last_relax_stats()
    # (a batch this small takes double tiles: 3.97 evaluations per 128 rows, counted as two each)
    assert tiles > 3000 and evals < 4.5 * tiles, ("evaluations per tile, double tiles", tiles, evals)
    assert slow <= 2 * 4, ("only the last tile of a topic leaves the straight-line path", slow)
    assert_same_outputs(fb, want, emu_solve(fb, flags=RELAX_TILES_64), "emu C3, relaxation form over tiles of 64 rows")
    tiles, evals, slow = last_relax_stats()
    assert tiles > 3000 and evals < 3.6 * tiles, ("evaluations per tile", tiles, evals)   # measured 3.26
    assert slow <= 2 * 4, ("only the last tile of a topic leaves the straight-line path", slow)
    assert_same_outputs(fb, want, emu_solve(fb, flags=TICKET_ORDER), "emu C3, ticket form")
    assert last_order_form() == 1 and last_queue_rows() > 1000, "the queue path of the 3-wide solver did not run"
    for flags, what in ((1, "general fill"), (0x200000, "quota drawn without the atomic-with-return"), (2, "round form"), (4, "4 x uint16 counter rows"), (8, "chunk-count pass"),
                        ((1 << 8) | (1 << 12), "1 fill wave, 1 scenario per wavefront"),
                        ((2 << 8) | (2 << 12), "2 fill waves, 2 scenarios per wavefront")):
        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"emu C3 {what}")


def test_emu_config4_exact_action_add_brokers_1000_to_1049():
    """configs[3]'s action at full size: add brokers 1000-1049 (rack id mod 20): N = 1050, cap 286, 14k-16k orphans."""
    fb = _batch(4004, 3, 100000, 1000, 20, 3, ("add50",))
    assert (fb.scen["n_nodes"] == 1050).all()
    want = oracle_solve(fb)
    assert (want.scenario_results["status"] == abi.KAS_OK).all()
    assert (want.scenario_results["moved_replicas"] > 10000).all()
    assert_same_outputs(fb, want, emu_solve(fb), "emu C4 add 50")
    assert last_order_form() == 3
    assert_same_outputs(fb, want, emu_solve(fb, flags=TICKET_ORDER), "emu C4 add 50, ticket form")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("rack_aware", [True, False])
def test_emu_config5_full_size_1m_partitions_5k_brokers_rf5(rack_aware):
    """configs[4] at its stated size: 1M partitions x 5k brokers x 40 racks, RF 5, remove every 50th broker + add
    5000-5199, rack map as generated / empty (--disable_rack_awareness): N = 5100, cap 981, ~219k moved replicas —
    the spread fill's scan kernels and the wide ticket form with its joint solve."""
    P, N, R, RF = 1000000, 5000, 40, 5
    cur = G.random_assignment(7, P, N, R, RF)
    bs = G.perturb_brokers(N, R, remove=list(range(0, N, 50)), add=200, rack_aware=rack_aware)
    assert bs.node_id.shape[0] == 5100
    fb = uniform_batch(cur[None], bs.node_id[None], bs.node_rack[None], RF)
    want = oracle_solve(fb)
    assert want.scenario_results["status"][0] == abi.KAS_OK and want.scenario_results["moved_replicas"][0] > 200000
    got = emu_solve(fb)
    assert last_order_form() == 2, "not the wide ticket form"
    assert last_spread() == 1, "not the spread fill"
    assert_same_outputs(fb, want, got, f"emu C5 full size rack_aware={rack_aware}")
