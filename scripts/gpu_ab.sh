#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # inflight tag
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu --check 4 --in-flight $1 --stats gpurun_out/stats_$2.json > gpurun_out/bench_$2.log 2>&1
  echo "exit $?" >> gpurun_out/bench_$2.log
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$2.log").read().strip().splitlines()[-2])
    r=d["roofline"]
    print("$2", "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "fill_us", round(r["fill_kernel_avg_us"]), "order_us", round(r["order_kernel_avg_us"]))
    st=json.load(open("gpurun_out/stats_$2.json"))
    print({k:(round(v["mean"],1),round(v["max"],1)) for k,v in st.items() if isinstance(v,dict) and k in ("order_us","p2_hist_quota_us","p2_keep_p3_us","p4_us","setup_us")})
except Exception as e:
    print("$2 FAILED", e); print(open("gpurun_out/bench_$2.log").read()[-1500:])
PY
}
run 1 new_f1
run 8 new_f8
export KAS_HIP_LIB=$PWD/kafka-assigner_amd/csrc/libkas_hip_prev.so
run 1 prev_f1
run 8 prev_f8
unset KAS_HIP_LIB
run 1 new_f1b
run 8 new_f8b
