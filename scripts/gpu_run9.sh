#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
run() {  # waves groups inflight tag
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --check 4 --waves $1 --groups $2 --in-flight $3 --stats gpurun_out/stats_$4.json > gpurun_out/bench_$4.log 2>&1
  echo "exit $?" >> gpurun_out/bench_$4.log
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$4.log").read().strip().splitlines()[-2])
    r=d["roofline"]
    print("$4", "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "fill_us", round(r["fill_kernel_avg_us"]), "order_us", round(r["order_kernel_avg_us"]))
    st=json.load(open("gpurun_out/stats_$4.json"))
    print({k:(round(v["mean"],1),round(v["max"],1)) for k,v in st.items() if isinstance(v,dict) and k in ("order_us","solver_iterations","stager_iterations","p2_hist_quota_us","p2_keep_p3_p4_us")})
except Exception as e:
    print("$4 FAILED", e); print(open("gpurun_out/bench_$4.log").read()[-1500:])
PY
}
run 4 1 1 f1
run 4 1 4 f4
run 4 1 8 f8
run 8 1 8 w8f8
run 4 2 8 g2f8
