#!/bin/bash
set -x
O=gpurun_out/r2j
mkdir -p $O
export TMPDIR=/tmp
export KAS_HIP_LIB=$PWD/variants/libkas_hip_fus.so
for pf in 0 8 0 8; do
  timeout 200 python bench.py --no-cpu --check 8 --no-extras --steps 40 --plan-flags $pf > $O/bench_pf$pf.log 2>&1; echo "exit $?" >> $O/bench_pf$pf.log
  echo "flags $pf $(tail -2 $O/bench_pf$pf.log | cut -c1-130)"
  timeout 200 python bench.py --no-cpu --check 0 --no-extras --steps 10 --in-flight 1 --plan-flags $pf --stats $O/stats1_pf$pf.json > $O/bench1_pf$pf.log 2>&1
  grep -o '"in_flight_launch": {[^}]*' $O/bench1_pf$pf.log | cut -c1-120
done
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 2 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/prof_fetch.log 2>&1; echo "fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_write -o write -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 2 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/prof_write.log 2>&1; echo "write exit $?"
