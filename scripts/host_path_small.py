"""Latency of kas_solve_host on small batches: new shapes (plan creation) and repeated shapes (cache)."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from test_emu_parity import _batch
from kafka_assigner_amd import native, generator as G
ctx = native.DeviceContext(0)
shapes = [(300 + 40 * i, 12 + i, 4, 3) for i in range(12)] + [(300, 12, 4, 5), (400, 14, 7, 4)]
fbs = [_batch(5 + i, 2, P, N, R, RF, ("add_k", "remove1")) for i, (P, N, R, RF) in enumerate(shapes)]
for rnd in range(3):
    ts = []
    for fb in fbs:
        t = time.perf_counter(); native.solve_host(fb, ctx); ts.append((time.perf_counter() - t) * 1e3)
    print(f"round {rnd}: per-call ms:", " ".join(f"{x:.1f}" for x in ts))
fb = fbs[0]
t = time.perf_counter()
for _ in range(50):
    native.solve_host(fb, ctx)
print(f"same shape x50: {(time.perf_counter() - t) * 20:.2f} ms per call; stats", ctx.host_stats())
