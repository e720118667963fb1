"""Randomised GPU-vs-oracle stress (not part of the pytest suites): random shapes (8..500 brokers,
300..20000 partitions, RF 2..3, every action mix, 1..8 scenarios per batch) for N seconds, each batch
solved with the default plan, one scenario per solver wavefront, 4 x uint16 counter rows, and the spread fill
forced, and compared bit for bit with the CPU oracle.  Usage: python scripts/stress_gpu.py SECONDS
(round 1: 21,494 batches x 3 plan variants in 150 s on an MI355X, all identical)."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from test_emu_parity import _batch
from oracle_lib import oracle_solve
from parity_util import assert_same_outputs
from kafka_assigner_amd import native, generator as G
rng = np.random.default_rng(2026)
t0 = time.time(); n = 0; q_rows = 0
while time.time() - t0 < float(sys.argv[1]):
    N = int(rng.choice([8, 12, 20, 33, 64, 100, 150, 300, 500]))
    R = int(rng.choice([2, 3, 5, 8, 10, 20])); R = min(R, N)
    RF = int(rng.choice([2, 3, 3, 3, 4, 5])); RF = min(RF, R)     # 4, 5: the wide ticket form
    P = int(rng.choice([300, 1000, 3000, 7000, 20000]))
    acts = [("add_k",), ("remove1",), ("remove_k", "mixed"), G.ACTIONS, ("mixed", "add_k")][int(rng.integers(5))]
    seed = int(rng.integers(1 << 30))
    S = int(rng.choice([1, 2, 3, 5, 8]))
    fb = _batch(seed, S, P, N, R, RF, acts)
    want = oracle_solve(fb)
    # (32 = KAS_PLAN_SPREAD_FILL: the row scans over one-wavefront workgroups with their slim LDS layouts)
    for flags in ((0, 1 << 12, 4, 32) if RF <= 3 else (0, 2, 1, 32)):
        got = native.solve_host_with_flags(fb, flags) if flags else native.solve_host(fb)
        assert_same_outputs(fb, want, got, f"seed {seed} S{S} P{P} N{N} R{R} RF{RF} {acts} flags {flags}")
    n += 1
print("stress ok:", n, "random batches x 4 plan variants")
