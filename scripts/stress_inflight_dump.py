"""Catch one wrong solve of the in-flight regime and look at it: per-broker replica counts against the
cap, first row that differs from the oracle.  (Diagnosis companion of stress_inflight.py.)

  python scripts/stress_inflight_dump.py [ROUNDS] [bench.py flags]
"""
import math
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

rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
args = bench.parse_args([a for a in sys.argv[2:]])
args.steps = max(args.steps, args.in_flight)
mix = tuple(args.actions.split(",")) if args.actions else G.BENCH_ACTIONS
run = bench.HipRun(args, 0, 1, 0, 0, args.scenarios, mix)
S, P, RF = run.S, args.partitions, args.rf
run.solve(run.slots[0]); run.synchronize()
ref_sr = run.slots[0]["sr"].clone()
run.synchronize()
found = 0
for r in range(rounds):
    for k in range(3):
        for sl in run.slots:
            run.solve(sl)
    run.synchronize()
    for i, sl in enumerate(run.slots):
        if torch.equal(sl["sr"], ref_sr):
            continue
        a = sl["sr"].cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE); b = ref_sr.cpu().numpy().view(abi.SCENARIO_RESULT_DTYPE)
        for s in np.nonzero(a != b)[0][:2]:
            s = int(s); found += 1
            sub = node_set_batch([run.ids[s]], [run.racks[s]], P, RF, RF, cur=run.host_cur([s]))
            want = oracle_solve(sub)
            w = want.out[:P * RF].reshape(P, RF)
            g = sl["out"].view(S, P, RF)[s].cpu().numpy()
            n_nodes = len(run.ids[s]); cap = math.ceil(P * RF / n_nodes)
            cg = np.bincount(g.reshape(-1), minlength=int(run.ids[s].max()) + 1); cw = np.bincount(w.reshape(-1), minlength=len(cg))
            over = np.nonzero(cg > cap)[0]
            d = np.nonzero((np.sort(g, axis=1) != np.sort(w, axis=1)).any(axis=1))[0]
            d_any = np.nonzero((g != w).any(axis=1))[0]
            print(f"round {r} slot {i} scenario {s} ({run.actions[s]}, {n_nodes} brokers, cap {cap}): got {a[s]} want {b[s]}")
            print(f"   brokers over cap in got: {[(int(x), int(cg[x])) for x in over[:8]]}; max count oracle {int(cw.max())}")
            print(f"   rows whose broker SET differs: {len(d)} (first {d[:5].tolist()}); rows that differ at all: {len(d_any)} (first {d_any[:3].tolist()})")
            for p_ in d[:3].tolist():
                print(f"     row {p_}: got {g[p_].tolist()} oracle {w[p_].tolist()}  counts got {[int(cg[x]) for x in g[p_]]} oracle {[int(cw[x]) for x in w[p_]]}")
            diff_b = np.nonzero(cg != cw)[0]
            print(f"   brokers whose replica count differs: {[(int(x), int(cg[x]), int(cw[x])) for x in diff_b[:12]]}")
    if found >= 3:
        break
print(f"{found} wrong scenario solves looked at")
