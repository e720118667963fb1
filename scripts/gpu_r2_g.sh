#!/bin/bash
# round 2, trip G: host path cache, batched JNI, watchdog build, bounded stress
set -x
O=gpurun_out/r2g
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
KAS_HIP_LIB=$PWD/variants/libkas_hip_bounded.so timeout 200 python scripts/stress_gpu.py 45 > $O/stress_bounded.log 2>&1; echo "stress exit $?" >> $O/stress_bounded.log; tail -2 $O/stress_bounded.log
KAS_HIP_LIB=$PWD/variants/libkas_hip_stall3.so timeout 90 python scripts/watchdog_gpu.py --narrow-only > $O/watchdog3.log 2>&1; echo "exit $?" >> $O/watchdog3.log; tail -3 $O/watchdog3.log
KAS_HIP_LIB=$PWD/variants/libkas_hip_stall5.so timeout 90 python - > $O/watchdog5.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_emu_parity import _batch
from oracle_lib import oracle_solve
from kafka_assigner_amd import abi, native
fb = _batch(78, 6, 30000, 300, 12, 5, ("add_k", "remove1"))
want = oracle_solve(fb); t = time.time(); got = native.solve_host(fb)
ok = want.scenario_results["status"] == abi.KAS_OK
assert ok.any() and (got.scenario_results["status"][ok] == abi.KAS_FAIL_WATCHDOG).all(), got.scenario_results["status"]
print(f"RF 5: stalled solve returned in {time.time() - t:.2f} s with KAS_FAIL_WATCHDOG on {int(ok.sum())} scenarios")
PY
echo "exit $?" >> $O/watchdog5.log; tail -3 $O/watchdog5.log
timeout 300 python scripts/host_path_rate.py 200 > $O/host_path_rate.log 2>&1; tail -2 $O/host_path_rate.log
KAS_HIP_LIB=$PWD/variants/libkas_hip_cur.so timeout 200 python bench.py --no-cpu --check 8 --no-extras --steps 40 > $O/bench_cur.log 2>&1; echo "cur $(tail -1 $O/bench_cur.log | cut -c1-130)"
for act in c5 c5_norack; do
  KAS_HIP_LIB=$PWD/variants/libkas_hip_w5cur.so timeout 300 python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions $act --in-flight 1 --steps 6 --warmup 1 > $O/bench_w5cur_$act.log 2>&1
  echo "$act $(grep -o '"in_flight_launch": {[^}]*' $O/bench_w5cur_$act.log | cut -c1-110)"
done
