#!/bin/bash
# scripts/gpu_c5v.sh OUTDIR VARIANT... : configs[4] (rack map on) once per tuning build, order-kernel time and steps
O=gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp
for v in "$@"; do
  KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout ${C5_TIMEOUT:-75} python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions ${C5_ACT:-c5} --in-flight 1 --steps 4 --warmup 1 --stats $O/stats_$v.json > $O/bench_$v.log 2>&1
  echo "C5 $v exit $? $(grep -o '"in_flight_launch": {[^}]*' $O/bench_$v.log | cut -c1-100) $(python -c "
import json; d=json.load(open('$O/stats_$v.json')); print('steps', round(d['solver_iterations']['mean']), 'queue rows', round(d['solver_queue_rows']['mean']), 'rounds', round(d['solver_queue_rounds']['mean']), 'bulk', round(d.get('p5_rounds_or_queue_steps',{}).get('mean',0)))" 2>&1 | tail -1)"
done
