#!/bin/bash
O=gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp
for v in "$@"; do
  for act in c5 c5_norack; do
    KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout ${C5_TIMEOUT:-120} python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions $act --in-flight 1 --steps 6 --warmup 1 --stats $O/stats_${v}_$act.json > $O/bench_${v}_$act.log 2>&1
    echo "C5 $v $act $(grep -o '"in_flight_launch": {[^}]*' $O/bench_${v}_$act.log | cut -c1-100) $(python -c "
import json; d=json.load(open('$O/stats_${v}_$act.json')); print('steps', round(d['solver_iterations']['mean']), 'queue rows', round(d['solver_queue_rows']['mean']))")"
  done
done
