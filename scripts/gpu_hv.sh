#!/bin/bash
# scripts/gpu_hv.sh OUTDIR VARIANT... : the headline bench (all scenarios checked) once per tuning build
O=gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp
for v in "$@"; do
  KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout ${HV_TIMEOUT:-150} python bench.py --no-cpu --no-extras --steps ${HV_STEPS:-20} --warmup 5 > $O/bench_$v.log 2>&1
  echo "HL $v exit $? $(tail -1 $O/bench_$v.log | cut -c1-160)"
done
