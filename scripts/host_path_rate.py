"""PCIe-inclusive rate of the host-buffer entry point (kas_solve_host): H2D of the tables, both
kernels, D2H of out + records, for a batch of S scenarios at 100k x 1k x RF 3.  Not the headline
metric (bench.py times with inputs resident in HBM); DESIGN.md section 8 quotes this number."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_emu_parity import _batch
from kafka_assigner_amd import native, generator as G
S = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fb = _batch(77, S, 100000, 1000, 20, 3, G.BENCH_ACTIONS)
native.solve_host(fb)                                  # warm-up (library load, first allocation)
t = time.perf_counter(); n = 3
for _ in range(n):
    native.solve_host(fb)
dt = (time.perf_counter() - t) / n
print(f"kas_solve_host: {S} scenarios in {dt * 1e3:.1f} ms = {S / dt:.0f} scenarios/s "
      f"({2 * S * 1.2e6 / dt / 1e9:.1f} GB/s of table traffic over PCIe)")
