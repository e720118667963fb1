#!/bin/bash
# round 2, second half: evidence run with the production library (parity tests, smoke, the bench line at
# the driver's command line and at the default, the BASELINE configs, the in-flight consistency stress,
# kernel traces, PMC traffic passes).  scripts/collect_profiles.py copies the summaries into profiles/.
set -x
O=gpurun_out/${1:-r2b}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmdline.log 2>&1; echo "exit $?" >> $O/bench_driver_cmdline.log
tail -2 $O/bench_driver_cmdline.log | cut -c1-200
timeout 900 python bench.py --no-cpu --stats $O/stats_default.json > $O/bench_default.log 2>&1; echo "exit $?" >> $O/bench_default.log
tail -2 $O/bench_default.log | cut -c1-200
timeout 300 python bench.py --no-cpu --check 0 --no-extras --steps 10 --in-flight 1 --stats $O/stats_one_batch_in_flight.json > $O/bench_one_batch_in_flight.log 2>&1
timeout 300 python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 10000 --brokers 100 --racks 10 --actions remove1 --in-flight 1 --steps 50 --warmup 5 > $O/bench_config2_single_scenario.log 2>&1
tail -1 $O/bench_config2_single_scenario.log | cut -c1-160
timeout 900 python bench.py --no-cpu --no-extras --check 64 --scenarios 8000 --actions add50 --in-flight 2 --steps 4 --warmup 1 > $O/bench_config4_8000_scenarios_add50.log 2>&1
tail -1 $O/bench_config4_8000_scenarios_add50.log | cut -c1-160
for act in c5 c5_norack; do
  timeout 300 python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions $act --in-flight 1 --steps 6 --warmup 1 --stats $O/stats_config5_$act.json > $O/bench_config5_$act.log 2>&1
  echo "$act $(grep -o '"ms_per_step": [0-9.]*' $O/bench_config5_$act.log) $(grep -o '"in_flight_launch": {[^}]*' $O/bench_config5_$act.log | cut -c1-110)"
done
timeout 300 python bench.py --no-cpu --no-extras --check 2 --scenarios 8 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 5 --warmup 1 > $O/bench_config5_x8.log 2>&1
echo "x8 $(grep -o '"ms_per_step": [0-9.]*' $O/bench_config5_x8.log)"
timeout 200 python scripts/stress_inflight.py 200 > $O/stress_inflight.log 2>&1; echo "stress exit $?" >> $O/stress_inflight.log; tail -3 $O/stress_inflight.log | cut -c1-200
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_trace_one_batch_in_flight -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --in-flight 1 --steps 20 > $GRAFT_REPO_ROOT/$O/prof_trace_f1.log 2>&1; echo "trace exit $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_trace_default -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras > $GRAFT_REPO_ROOT/$O/prof_trace_default.log 2>&1; echo "trace exit $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_trace_config5 -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --check 0 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 6 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof_trace_c5.log 2>&1; echo "trace exit $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 2 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/prof_fetch.log 2>&1; echo "fetch exit $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_write -o write -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 2 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/prof_write.log 2>&1; echo "write exit $?"
cd $GRAFT_REPO_ROOT
