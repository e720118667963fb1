#!/bin/bash
set -x
O=gpurun_out/r2i
mkdir -p $O
export TMPDIR=/tmp
for v in pf1 exp1 exp2 pf1 exp1 exp2; do
  KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout 200 python bench.py --no-cpu --check 0 --no-extras --steps 40 > $O/bench_$v.log 2>&1; echo "exit $?" >> $O/bench_$v.log
  echo "$v $(tail -2 $O/bench_$v.log | cut -c1-130)"
  KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout 200 python bench.py --no-cpu --check 0 --no-extras --steps 10 --in-flight 1 --stats $O/stats1_$v.json > $O/bench1_$v.log 2>&1
  grep -o '"in_flight_launch": {[^}]*' $O/bench1_$v.log | cut -c1-120
done
