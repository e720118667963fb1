#!/bin/bash
export TMPDIR=/tmp
export KAS_HIP_LIB=$PWD/variants/libkas_hip_claim.so
for cfg in "8 16 20 5" "12 16 20 5" "12 24 20 5" "14 16 20 5" "12 16 40 8" "12 24 40 8" "10 16 20 5" "12 16 20 12"; do
  set -- $cfg
  for rep in 1 2; do
    V=$(GPU_MAX_HW_QUEUES=$2 timeout 200 python bench.py --no-cpu --check 0 --no-extras --steps $3 --warmup $4 --in-flight $1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']))")
    echo "CFG inflight=$1 hwq=$2 steps=$3 warmup=$4 rep$rep $V"
  done
done
