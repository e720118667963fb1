"""GPU check of the hang containment build (run with KAS_HIP_LIB pointing at a variant built with
-DKAS_SPIN_BOUND=n -DKAS_TEST_STALL_AFTER=k): the staging wavefront stops handing out rows, the
solve must still RETURN and report KAS_FAIL_WATCHDOG.  Wrap in `timeout`."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from test_emu_parity import _batch
from oracle_lib import oracle_solve
from kafka_assigner_amd import abi, native, generator as G
for RF in (3,) if "--narrow-only" in sys.argv else (3, 5):
    fb = _batch(78, 6, 30000, 300, 12, RF, ("add_k", "remove1"))
    want = oracle_solve(fb)
    t = time.time()
    got = native.solve_host(fb)
    ok = want.scenario_results["status"] == abi.KAS_OK
    assert ok.any()
    assert (got.scenario_results["status"][ok] == abi.KAS_FAIL_WATCHDOG).all(), got.scenario_results["status"]
    assert (got.scenario_results["status"][~ok] == want.scenario_results["status"][~ok]).all()
    print(f"RF {RF}: stalled solve returned in {time.time() - t:.2f} s with KAS_FAIL_WATCHDOG on {int(ok.sum())} scenarios")
print("watchdog ok")
