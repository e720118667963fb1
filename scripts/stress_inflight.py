"""Consistency stress of the headline regime: every slot of bench.py solves the SAME inputs, so all
solves must leave identical scenario records (status, movement, digest of every emitted cell).
Issues ROUNDS x slots solves back to back on the slots' streams (no host synchronisation in between,
as bench.py's timed region does) and counts, on each slot's own stream right behind its solve, the
scenarios whose record differs from a reference solve that ran alone and is checked against the oracle.

  python scripts/stress_inflight.py [ROUNDS] [bench.py flags, e.g. --waves 1 --in-flight 12]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from kafka_assigner_amd import abi, generator as G  # noqa: E402
from kafka_assigner_amd.flatten import node_set_batch  # noqa: E402
from oracle_lib import oracle_solve  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
args = bench.parse_args([a for a in sys.argv[2:]])
args.steps = max(args.steps, args.in_flight)
mix = tuple(args.actions.split(",")) if args.actions else G.BENCH_ACTIONS
run = bench.HipRun(args, 0, 1, 0, 0, args.scenarios, mix)
S, P, RF = run.S, args.partitions, args.rf
run.solve(run.slots[0]); run.synchronize()
ref_sr = run.slots[0]["sr"].clone()
run.synchronize()                      # (the copy runs on torch's stream, the next solves on the slots' own)
# the reference itself against the oracle (records of every scenario)
sub = node_set_batch(run.ids, run.racks, P, RF, RF, cur=run.host_cur(list(range(S))))
want = oracle_solve(sub, threads=0)
got = ref_sr.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
for f in ("status", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
    assert (got[f] == want.scenario_results[f][:S]).all(), f"the reference solve differs from the oracle in {f}"
ref_rec = ref_sr.view(S, 32)
counts = [torch.zeros(S, dtype=torch.int32, device=run.dev) for _ in run.slots]
for r in range(rounds):
    for i, sl in enumerate(run.slots):
        run.solve(sl)
        with run.stream_ctx(sl):
            counts[i] += (sl["sr"].view(S, 32) != ref_rec).any(dim=1).to(torch.int32)
run.synchronize()
tot = torch.stack(counts).sum(dim=0).cpu().numpy()
bad = np.nonzero(tot)[0]
n = rounds * len(run.slots)
print(f"{n} solves of {S} scenarios, {len(run.slots)} in flight: {int(tot.sum())} scenario records differ from the reference"
      + (f" (scenarios {bad[:10].tolist()}, actions {[run.actions[s] for s in bad[:10]]})" if len(bad) else ""))
print("plan:", run.describe())
sys.exit(1 if len(bad) else 0)
