#!/bin/bash
# A/B of tuning builds on configs[2]: scripts/gpu_ab.sh OUTDIR variant...   (each variant twice, alternating)
O=gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout 200 python bench.py --no-cpu --check 0 --no-extras --steps 40 > $O/bench_${v}_$rep.log 2>&1
  A=$(tail -1 $O/bench_${v}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']))" 2>/dev/null)
  KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout 200 python bench.py --no-cpu --check 0 --no-extras --steps 10 --in-flight 1 --stats $O/stats1_${v}_$rep.json > $O/bench1_${v}_$rep.log 2>&1
  B=$(tail -1 $O/bench1_${v}_$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']['in_flight_launch']; print(round(r['fill_kernel_us']), round(r['order_kernel_us']))" 2>/dev/null)
  echo "AB $v rep$rep in-flight-8: $A scen/s; alone fill/order us: $B"
done
done
