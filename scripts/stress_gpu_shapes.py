"""GPU-vs-oracle stress over the plan's shape decisions (not part of the pytest suites): broker counts
up to 6000 and list widths 2..6, so that every LDS carve-up is taken at least once — per-chunk
histograms / one histogram + chunk-count pass / general fill only, 4 / 2 / 1 fill waves, direct id
table / binary search (sparse ids), ticket form 3-wide / wide / round form — plus every plan switch
that applies.  Usage: python scripts/stress_gpu_shapes.py SECONDS"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from test_emu_parity import _batch
from oracle_lib import oracle_solve
from parity_util import assert_same_outputs
from kafka_assigner_amd import native, generator as G
rng = np.random.default_rng(77)
t0 = time.time(); n = 0; seen = {}
shapes = [(1000, 20, 3), (2000, 40, 3), (4000, 40, 3), (6000, 60, 3), (3000, 30, 2), (1500, 30, 4), (2500, 50, 5),
          (5100, 40, 5), (900, 30, 6), (6000, 60, 5), (700, 14, 3)]
while time.time() - t0 < float(sys.argv[1]):
    N, R, RF = shapes[n % len(shapes)]
    P = int(rng.choice([20000, 60000, 150000]))
    S = int(rng.choice([1, 2, 3]))
    acts = [("add_k",), ("remove1", "mixed"), G.ACTIONS][int(rng.integers(3))]
    seed = int(rng.integers(1 << 30))
    fb = _batch(seed, S, P, N, R, RF, acts)
    if n % 5 == 4:                                     # sparse ids: binary search instead of the direct table
        fb.node_id = (fb.node_id.astype(np.int64) * 7919 + 3).astype(np.int32)
        fb.cur = np.where(fb.cur >= 0, fb.cur.astype(np.int64) * 7919 + 3, -1).astype(np.int32)
    want = oracle_solve(fb, threads=0)
    ctx = native.default_context()
    plan = native.Plan(ctx, fb); desc = plan.describe(); plan.close()
    seen[desc.split(" grid")[0] + " | " + desc.split("+ ")[-1].split(" grid")[0]] = seen.get(desc, 0) + 1
    assert_same_outputs(fb, want, native.solve_host(fb), f"seed {seed} S{S} P{P} N{N} R{R} RF{RF} {acts}")
    for flags in ((1, 2, 8) if RF <= 3 else (1, 2)):
        if flags == 2 and 28 * fb.scen["n_nodes"].max() > 160 * 1024 and RF == 5:
            continue                                   # the round form's LDS does not fit: the plan refuses the switch
        assert_same_outputs(fb, want, native.solve_host_with_flags(fb, flags), f"seed {seed} N{N} RF{RF} flags {flags}")
    n += 1
print("stress (shapes) ok:", n, "batches;", len(seen), "kernel combinations:")
for k in sorted(seen): print("  ", k)
