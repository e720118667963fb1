#!/usr/bin/env python
"""The plain host path in seconds (MEASUREMENT TOOLING; bench.py's end_to_end leg holds the figures that are quoted):
S scenarios with their own 100,000 x 3 tables through kas_solve_host (int32 broker ids) and kas_solve_host16 (16-bit
node indices), caller buffers from kas_host_alloc; and the number of scenario ranges a call is cut into
(KAS_HOST_RANGES, a tuning override of kas_solve_host's own choice).  usage: e2e_host_path.py [S] [calls]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kafka_assigner_amd import generator as G, native  # noqa: E402
from kafka_assigner_amd.flatten import (batch_desc, cells16_to_ids, host_tables, host_tables16, node_set_batch,  # noqa: E402
                                        to_cells16)

S = int(sys.argv[1]) if len(sys.argv) > 1 else 240
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
P, N, R = 100000, 1000, 10
base = [G.random_assignment(50 + k, P, N, R, 3) for k in range(4)]
cur = np.stack([base[s % 4] for s in range(S)])
sets = [G.scenario_action(9, s, N, R, actions=G.BENCH_ACTIONS, max_add=50)[1] for s in range(S)]
fb = node_set_batch([b.node_id for b in sets], [b.node_rack for b in sets], P, 3, 3, cur=cur)
L = native.load()
ctx = native.DeviceContext(0)
bd = batch_desc(fb)
pc, po = native.PinnedArray(fb.cur.size), native.PinnedArray(fb.out_len)
pc.array[:] = fb.cur; po.array[:] = 0
t, ho = host_tables(fb)
t.cur = pc.array.ctypes.data; t.out = po.array.ctypes.data
c16 = to_cells16(fb)
pc16, po16 = native.PinnedArray(c16.size, np.uint16), native.PinnedArray(fb.out_len, np.uint16)
pc16.array[:] = c16; po16.array[:] = 0
t16, ho16 = host_tables16(fb, pc16.array)
t16.out = po16.array.ctypes.data
bd16 = batch_desc(fb)
bd16.node_id = None


def run(name, fn, cell):
    for _ in range(2):
        fn()
    t0 = time.perf_counter()
    for _ in range(calls):
        fn()
    dt = (time.perf_counter() - t0) / calls
    print(f"{name:58s} {1e3 * dt:7.2f} ms per call  {S / dt / 1e3:6.1f}k scenarios/s  {cell * (fb.cur.size + fb.out_len) / dt / 1e9:5.1f} GB/s over the link (sum)", flush=True)


f32 = lambda: native._check(L.kas_solve_host(ctx._h, C.byref(bd), C.byref(t)))
f16 = lambda: native._check(L.kas_solve_host16(ctx._h, C.byref(bd16), C.byref(t16), None, -1))
run("int32 broker ids, pinned", f32, 4)
ids32 = po.array.copy()
run("16-bit cells, pinned", f16, 2)
assert (cells16_to_ids(fb, po16.array) == ids32).all(), "16-bit rows differ from the int32 call's"
for k in (os.environ.get("E2E_RANGES", "2,3,4,6,8").split(",")):          # scenario ranges per call (KAS_HOST_RANGES: tuning)
    os.environ["KAS_HOST_RANGES"] = k
    run(f"int32, {k} scenario ranges", f32, 4)
    run(f"16-bit cells, {k} scenario ranges", f16, 2)
del os.environ["KAS_HOST_RANGES"]
print("rows identical")
