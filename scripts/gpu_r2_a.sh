#!/bin/bash
# round 2, trip A: driver-style evidence with the new harness + first look at the experiment variants
set -x
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
export TMPDIR=/tmp
nproc > $O/nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -14 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 900 python bench.py --stats $O/stats_default.json > $O/bench_default.log 2>&1; echo "exit $?" >> $O/bench_default.log
tail -2 $O/bench_default.log | cut -c1-1500
timeout 120 python bench.py --gpus 2 --steps 4 > $O/bench_gpus2.log 2>&1; echo "exit $?" >> $O/bench_gpus2.log; tail -3 $O/bench_gpus2.log
for v in base handover tw handover_tw8; do
  KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout 150 python bench.py --no-cpu --check 8 --no-extras --steps 40 > $O/bench_v_$v.log 2>&1; echo "exit $?" >> $O/bench_v_$v.log
  tail -2 $O/bench_v_$v.log | cut -c1-260
done
# kernel trace with ONE batch in flight (per-kernel durations free of other launches)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_trace_f1 -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --in-flight 1 --steps 20 > $GRAFT_REPO_ROOT/$O/prof_trace_f1.log 2>&1; echo "trace exit $?"
cd $GRAFT_REPO_ROOT
ls -R $O | head -40
