"""scripts/whatif_handback_probe.py [S]: what-if calls (kas_solve_host_select: S broker-set variants over ONE snapshot, fresh sets on
every call) on a snapshot whose rows are NOT rack-diverse for the variants' rack map (racks assigned id mod 10 where the snapshot
was laid out over 20): every scenario is handed back by the slim fill kernel, and the plan of the host path is rebuilt in place
for every new set of variants.  Prints the time of each call and checks the records of the last one against the CPU solver.
MEASUREMENT TOOLING (GPU only)."""
import ctypes as C
import sys
import time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from kafka_assigner_amd import generator as G, native
from kafka_assigner_amd.flatten import batch_desc, host_tables, node_set_batch
from oracle_lib import cpu_fast_solve

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
P, N, R, RF = 100000, 1000, 20, 3
snapshot = G.random_assignment(11, P, N, R, RF)
L = native.load()
ctx = native.DeviceContext(0)


def variants(seed, diverse):
    ids, racks = [], []
    for s in range(S):
        _, bs = G.scenario_action(seed, s, N, R, actions=("remove1", "add_k"))
        ids.append(bs.node_id)
        racks.append(bs.node_rack if diverse else (bs.node_id % 10).astype(np.int32))
    return node_set_batch(ids, racks, P, RF, RF, shared_cur=True, cur=snapshot)


select = np.asarray([S // 2], dtype=np.int32)
for diverse in (True, False):
    times = []
    for i in range(6):
        fb = variants(100 + i, diverse)
        t, ho = host_tables(fb, out_len=native.selected_out_len(fb, select))
        bd = batch_desc(fb)
        t0 = time.perf_counter()
        native._check(L.kas_solve_host_select(ctx._h, C.byref(bd), C.byref(t), select.ctypes.data_as(C.POINTER(C.c_int32)), 1))
        times.append(1e3 * (time.perf_counter() - t0))
    want = cpu_fast_solve(fb, threads=0)
    for f in ("status", "fail_partition", "moved_replicas", "moved_partitions", "digest"):
        assert (ho.scenario_results[f][:S] == want.scenario_results[f][:S]).all(), f
    print(("rack-diverse snapshot" if diverse else "snapshot NOT rack-diverse (every variant handed back)") +
          f": {S} fresh variants per call, ms per call: " + " ".join("%.2f" % x for x in times) + "; records equal to the CPU solver's")
