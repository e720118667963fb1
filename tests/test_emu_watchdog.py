"""Hang containment (KAS_SPIN_BOUND; the product build bounds at 2^25 polls, these tests at far fewer): every loop in which the wavefronts of a
workgroup wait for each other is bounded; a wavefront that polls n times without progress raises
the workgroup's watchdog word, all polling loops leave, and the scenario reports
KAS_FAIL_WATCHDOG instead of hanging the GPU.  Checked on the CPU emulator of the kernel source:
(1) with the bound on and nothing wrong, results are the oracle's (no false alarm);
(2) with the staging wavefront made to stop handing out rows (test hook), the kernel RETURNS and
    says so — for the 3-wide and the wide ticket kernels."""
import numpy as np

from kafka_assigner_amd import abi
from kafka_assigner_amd import generator as G
from emu_lib import TICKET_ORDER, variant_solver
from oracle_lib import oracle_solve
from parity_util import assert_same_outputs
from test_emu_parity import _batch


def test_bounded_build_without_a_fault_equals_the_oracle():
    solve = variant_solver("bounded", ["-DKAS_SPIN_BOUND=200000"])
    for P, N, R, RF, acts in ((3000, 60, 6, 3, G.ACTIONS), (2000, 80, 12, 5, G.ACTIONS)):
        fb = _batch(77, 3, P, N, R, RF, acts)
        want = oracle_solve(fb)
        assert_same_outputs(fb, want, solve(fb), "bounded build")
        assert_same_outputs(fb, want, solve(fb, flags=TICKET_ORDER), "bounded build, ticket form")
        assert not (want.scenario_results["status"] == abi.KAS_FAIL_WATCHDOG).any()


def test_a_stalled_staging_wavefront_is_reported_not_hung():
    solve = variant_solver("stalled", ["-DKAS_SPIN_BOUND=1500", "-DKAS_TEST_STALL_AFTER=2"])
    for RF in (3, 5):                                   # order_tickets<3, ...> and order_tickets_wide<5>
        fb = _batch(78, 3, 1500, 120, 12, RF, ("add_k", "remove1"))
        want = oracle_solve(fb)
        got = solve(fb, flags=TICKET_ORDER)              # returns: that is the point (the relaxation form has one
                                                         # wavefront per scenario and nobody to wait for)
        ok = want.scenario_results["status"] == abi.KAS_OK
        assert ok.any()
        np.testing.assert_array_equal(got.scenario_results["status"][ok], abi.KAS_FAIL_WATCHDOG)
        # scenarios that had already failed in the fill kernel never reach the order kernel's rows
        np.testing.assert_array_equal(got.scenario_results["status"][~ok], want.scenario_results["status"][~ok])


def test_a_large_bound_is_checked_on_every_4096th_idle_poll_and_still_ends_the_solve():
    """The product's form of the check (bounds of 65536 and more): the watchdog word is read on every
    KAS_SPIN_CHECK-th poll without progress only.  Same stalled staging wavefront, smallest such bound."""
    solve = variant_solver("stalled_sparse", ["-DKAS_SPIN_BOUND=65536", "-DKAS_SPIN_CHECK=4096", "-DKAS_TEST_STALL_AFTER=2"])
    fb = _batch(78, 2, 1500, 120, 12, 3, ("add_k", "remove1"))
    want = oracle_solve(fb)
    ok = want.scenario_results["status"] == abi.KAS_OK
    assert ok.any()
    got = solve(fb, flags=TICKET_ORDER)
    np.testing.assert_array_equal(got.scenario_results["status"][ok], abi.KAS_FAIL_WATCHDOG)
