"""Step-count proxy for the wide ticket-form order kernel: a scaled-down BASELINE.json configs[4]
(same cap ~981 per broker: P x RF / N held, 1/10 of the brokers) through the CPU fiber emulator with
KAS_EMU_STATS=1, checked against the oracle.  The emulator is not a timing model; what it gives is the
number of solver steps / queue passes / rows decided in queues, which is what the class-1 chain costs
on hardware.  Test infrastructure.

  KAS_EMU_STATS=1 python scripts/emu_wide_steps.py [P] [N] [--norack]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("KAS_EMU_STATS", "1")

import numpy as np  # noqa: E402

from kafka_assigner_amd import generator as G  # noqa: E402
from kafka_assigner_amd.flatten import node_set_batch  # noqa: E402
from emu_lib import emu_solve, variant_solver  # noqa: E402
from oracle_lib import oracle_solve  # noqa: E402
from parity_util import assert_same_outputs  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
P = int(args[0]) if len(args) > 0 else 100000
N = int(args[1]) if len(args) > 1 else 500
act = "c5_norack" if "--norack" in sys.argv else "c5"
cur = G.random_assignment(11, P, N, 40, 5)[None]
bs = G.scenario_action(11, 0, N, 40, actions=(act,))[1]
fb = node_set_batch([bs.node_id], [bs.node_rack], P, 5, 5, cur=cur)
want = oracle_solve(fb)
print("oracle status", want.scenario_results["status"].tolist(), "moved", want.scenario_results["moved_replicas"].tolist())
t0 = time.time()
flags = os.environ.get("KAS_EMU_CFLAGS", "").split()     # e.g. "-DKAS_WIDE_DIAG -DKAS_WIDE_VOTE_DEPTH=2"
solve = variant_solver("v" + "".join(c for c in "_".join(flags) if c.isalnum() or c == "_"), flags) if flags else emu_solve
got = solve(fb)
print("emu %.1f s" % (time.time() - t0))
assert_same_outputs(fb, want, got, "emu wide proxy")
print("identical to the oracle")
