"""Randomised emulator-vs-oracle stress on CPU (test infrastructure; not part of the pytest suites):
random shapes and action mixes for N seconds, the kernel source stepped on the fiber emulator
(optionally with KAS_EMU_CHAOS=<seed> in the environment), every batch compared bit for bit with
the oracle for three plan variants.  Usage: [KAS_EMU_CHAOS=3] python scripts/stress_emu.py SECONDS"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from test_emu_parity import _batch
from emu_lib import emu_solve, last_queue_rows
from oracle_lib import oracle_solve
from parity_util import assert_same_outputs
from kafka_assigner_amd import generator as G
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0 = time.time(); n = 0; q = 0
while time.time() - t0 < float(sys.argv[1]):
    N = int(rng.choice([8, 12, 20, 33, 64, 100, 150]))
    R = min(int(rng.choice([2, 3, 5, 8, 10])), N)
    RF = min(int(rng.choice([2, 3, 3, 3])), R)
    P = int(rng.choice([200, 700, 1500, 4000, 9000]))
    acts = [("add_k",), ("remove1",), ("remove_k", "mixed"), G.ACTIONS, ("mixed", "add_k"), ("replace1",)][int(rng.integers(6))]
    seed = int(rng.integers(1 << 30)); S = int(rng.choice([1, 2, 3, 4]))
    fb = _batch(seed, S, P, N, R, RF, acts)
    want = oracle_solve(fb)
    for flags in (0, (1 << 12) | (2 << 8), 4 | (4 << 12) | (8 << 8)):
        assert_same_outputs(fb, want, emu_solve(fb, flags=flags), f"seed {seed} S{S} P{P} N{N} R{R} RF{RF} {acts} flags {flags:#x}")
        q += last_queue_rows()
    n += 1
print("emu stress ok:", n, "random batches x 3 plan variants,", q, "rows decided inside queues")
