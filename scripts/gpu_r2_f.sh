#!/bin/bash
# round 2, trip F: full library after the mid-row / register fixes: tests, bench, PMC traffic, trace
set -x
O=gpurun_out/r2f
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
for v in orig cur; do
  KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout 200 python bench.py --no-cpu --check 8 --no-extras --steps 40 > $O/bench_$v.log 2>&1; echo "exit $?" >> $O/bench_$v.log
  echo "$v $(tail -2 $O/bench_$v.log | cut -c1-130)"
done
timeout 900 python bench.py --stats $O/stats_default.json > $O/bench_default.log 2>&1; echo "exit $?" >> $O/bench_default.log
tail -2 $O/bench_default.log | cut -c1-200
grep -o '"one_batch_alone": {[^}]*' $O/bench_default.log
timeout 300 python bench.py --no-cpu --check 0 --no-extras --steps 10 --in-flight 1 --stats $O/stats_alone.json > $O/bench_alone.log 2>&1
for act in c5 c5_norack; do
  timeout 600 python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions $act --in-flight 1 --steps 6 --warmup 1 --stats $O/stats_$act.json > $O/bench_$act.log 2>&1; echo "exit $?" >> $O/bench_$act.log
  echo "$act $(grep -o '"in_flight_launch": {[^}]*' $O/bench_$act.log | cut -c1-110)"
done
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 2 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/prof_fetch.log 2>&1; echo "fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_write -o write -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 2 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/prof_write.log 2>&1; echo "write exit $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_trace_one_batch_in_flight -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --in-flight 1 --steps 20 > $GRAFT_REPO_ROOT/$O/prof_trace_f1.log 2>&1; echo "trace exit $?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_trace_default -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras > $GRAFT_REPO_ROOT/$O/prof_trace_default.log 2>&1; echo "trace exit $?"
cd $GRAFT_REPO_ROOT
