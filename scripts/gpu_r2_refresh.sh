#!/bin/bash
# after a change of the wide (lists 4, 5 wide) order kernel only: GPU tests, smoke, the driver's bench
# command line, BASELINE configs[4] (one scenario rack map on / off, one GPU's eight), shape stress
O=gpurun_out/${1:-r2refresh}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmdline.log 2>&1; echo "exit $?" >> $O/bench_driver_cmdline.log; tail -2 $O/bench_driver_cmdline.log | cut -c1-220
for act in c5 c5_norack; do
  timeout 120 python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions $act --in-flight 1 --steps 6 --warmup 1 --stats $O/stats_config5_$act.json > $O/bench_config5_$act.log 2>&1
  echo "$act $(grep -o '"ms_per_step": [0-9.]*' $O/bench_config5_$act.log) $(grep -o '"in_flight_launch": {[^}]*' $O/bench_config5_$act.log | cut -c1-110)"
done
timeout 200 python bench.py --no-cpu --no-extras --check 2 --scenarios 8 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 5 --warmup 1 > $O/bench_config5_x8.log 2>&1
echo "x8 $(grep -o '"ms_per_step": [0-9.]*' $O/bench_config5_x8.log)"
timeout 150 python scripts/stress_gpu_shapes.py 60 > $O/stress_shapes.log 2>&1; echo "stress exit $?" >> $O/stress_shapes.log; grep "stress" $O/stress_shapes.log
