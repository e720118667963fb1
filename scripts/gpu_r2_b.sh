#!/bin/bash
# round 2, trip B: 16-bit mid rows (traffic) + the wide ticket kernel (configs[4])
set -x
O=gpurun_out/r2b
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -10 $O/pytest_gpu.log
timeout 200 python scripts/stress_gpu.py 60 > $O/stress.log 2>&1; echo "stress exit $?" >> $O/stress.log; tail -2 $O/stress.log
timeout 900 python bench.py --no-cpu --stats $O/stats_default.json > $O/bench_default.log 2>&1; echo "exit $?" >> $O/bench_default.log
tail -2 $O/bench_default.log | cut -c1-300
# configs[4]: one scenario of 1M x 5k x RF 5, rack map on / off; then 8 scenarios (one GPU's share)
for act in c5 c5_norack; do
  timeout 600 python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions $act --in-flight 1 --steps 10 --warmup 2 --stats $O/stats_$act.json > $O/bench_$act.log 2>&1; echo "exit $?" >> $O/bench_$act.log
  tail -2 $O/bench_$act.log | cut -c1-2500
done
timeout 600 python bench.py --no-cpu --no-extras --check 2 --scenarios 8 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 5 --warmup 1 > $O/bench_c5x8.log 2>&1; echo "exit $?" >> $O/bench_c5x8.log
tail -2 $O/bench_c5x8.log | cut -c1-600
# the round form on the same scenario, for comparison (plan flag 2)
timeout 600 python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 3 --warmup 1 --plan-flags 2 > $O/bench_c5_round.log 2>&1
tail -1 $O/bench_c5_round.log | cut -c1-400
# PMC passes (own runs, one counter each, one batch in flight)
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 2 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/prof_fetch.log 2>&1; echo "fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_write -o write -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 2 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/prof_write.log 2>&1; echo "write exit $?"
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" | head; du -sh $O
