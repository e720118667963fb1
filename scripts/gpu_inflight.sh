#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
export KAS_HIP_LIB=$PWD/variants/libkas_hip_claim.so
for f in 2 4 6 8 10 12 16; do
  for rep in 1 2; do
    V=$(timeout 200 python bench.py --no-cpu --check 0 --no-extras --steps 48 --warmup 16 --in-flight $f | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']))")
    echo "INFLIGHT $f rep$rep $V"
  done
done
for q in 8 32; do
  V=$(GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --no-cpu --check 0 --no-extras --steps 48 --warmup 16 --in-flight 8 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']))")
  echo "HWQ $q $V"
done
