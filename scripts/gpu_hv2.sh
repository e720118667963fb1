#!/bin/bash
# scripts/gpu_hv2.sh OUTDIR REPS VARIANT... : the headline bench with its extra legs, REPS times per library ("full" = the in-tree build)
O=gpurun_out/$1; shift
R=$1; shift
mkdir -p $O
export TMPDIR=/tmp
for v in "$@"; do
  for i in $(seq 1 $R); do
    if [ "$v" == "full" ]; then unset KAS_HIP_LIB; else export KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so; fi
    timeout ${HV_TIMEOUT:-200} python bench.py --no-cpu --steps ${HV_STEPS:-20} --warmup 5 > $O/bench_${v}_$i.log 2>&1
    echo "HL $v #$i exit $? $(tail -1 $O/bench_${v}_$i.log | cut -c1-140)"
  done
done
