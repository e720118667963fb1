"""Step-count proxy for the 3-wide ticket-form order kernel at the headline shape (BASELINE.json
configs[2]: 100k x 1k x RF 3, the bench's action mix) through the CPU fiber emulator with
KAS_EMU_STATS=1, checked against the oracle.  The emulator is no timing model; what it gives is the
number of solver steps / queue passes / rows decided in queues.  Test infrastructure.

  python scripts/emu_ticket_steps.py [S] [P] [N]        KAS_EMU_CFLAGS="-D..." builds a variant
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
S = int(args[0]) if len(args) > 0 else 4
P = int(args[1]) if len(args) > 1 else 100000
N = int(args[2]) if len(args) > 2 else 1000
R = 20
cur = np.stack([G.random_assignment(100 + s, P, N, R, 3) for s in range(S)])
acts = G.BENCH_ACTIONS
sets = [G.scenario_action(7, s, N, R, actions=acts)[1] for s in range(S)]
fb = node_set_batch([b.node_id for b in sets], [b.node_rack for b in sets], P, 3, 3, cur=cur)
want = oracle_solve(fb)
print("oracle status", want.scenario_results["status"].tolist(), "moved", want.scenario_results["moved_replicas"].tolist())
t0 = time.time()
flags = os.environ.get("KAS_EMU_CFLAGS", "").split()
solve = variant_solver("v" + "".join(c for c in "_".join(flags) if c.isalnum() or c == "_"), flags) if flags else emu_solve
got = solve(fb)
print("emu %.1f s" % (time.time() - t0))
assert_same_outputs(fb, want, got, "emu ticket proxy")
print("identical to the oracle")
