#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # inflight tag
  timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu --check 4 --in-flight $1 > gpurun_out/bench_$2.log 2>&1
  echo "exit $?" >> gpurun_out/bench_$2.log
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$2.log").read().strip().splitlines()[-2])
    r=d["roofline"]
    print("$2", "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "fill_us", round(r["fill_kernel_avg_us"]), "order_us", round(r["order_kernel_avg_us"]))
except Exception as e:
    print("$2 FAILED", e); print(open("gpurun_out/bench_$2.log").read()[-1500:])
PY
}
run 8 q4_f8
export GPU_MAX_HW_QUEUES=8
run 8 q8_f8
run 12 q8_f12
export GPU_MAX_HW_QUEUES=16
run 8 q16_f8
run 12 q16_f12
run 16 q16_f16
export GPU_MAX_HW_QUEUES=24
run 16 q24_f16
run 24 q24_f24
