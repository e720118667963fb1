#!/usr/bin/env python
"""scripts/host_path_probe.py [pinned|pageable] [scenarios]: a few plain kas_solve_host calls (every scenario its own
100k x 3 tables) for a copy / kernel timeline under `rocprofv3 --kernel-trace --memory-copy-trace`.  MEASUREMENT TOOLING."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import conftest  # noqa: F401,E402
import numpy as np  # noqa: E402
from kafka_assigner_amd import generator as G, native  # noqa: E402
from kafka_assigner_amd.flatten import batch_desc, host_tables, node_set_batch  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "pinned"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 240
P, N, R, RF = 100000, 1000, 20, 3
ids, racks, curs = [], [], []
for s in range(m):
    _, bs = G.scenario_action(7, s, N, R, actions=G.BENCH_ACTIONS)
    ids.append(bs.node_id); racks.append(bs.node_rack)
cur = np.stack([G.random_assignment(100 + (s % 8), P, N, R, RF) for s in range(8)])
cur = cur[np.arange(m) % 8]
fb = node_set_batch(ids, racks, P, RF, RF, cur=cur)
L = native.load()
ctx = native.DeviceContext(0)
bd = batch_desc(fb)
t, ho = host_tables(fb)
def calls(what, t):
    for i in range(4):
        t0 = time.perf_counter()
        native._check(L.kas_solve_host(ctx._h, C.byref(bd), C.byref(t)))
        print("%s call %d: %.2f ms" % (what, i, 1e3 * (time.perf_counter() - t0)), flush=True)


if mode in ("pageable", "both"):
    calls("pageable", t)
if mode in ("pinned", "both"):
    if "torch" in sys.argv:
        import torch
        x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
    pc, po = native.PinnedArray(fb.cur.size), native.PinnedArray(fb.out_len)
    pc.array[:] = fb.cur
    t2, ho2 = host_tables(fb)
    t2.cur = pc.array.ctypes.data; t2.out = po.array.ctypes.data
    calls("pinned", t2)
