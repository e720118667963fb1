#!/bin/bash
# consolidated run: parity tests, bench sweeps, default bench line, kernel trace, PMC passes
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
run() {  # waves groups inflight tag
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --check 2 --waves $1 --groups $2 --in-flight $3 --stats gpurun_out/stats_$4.json > gpurun_out/bench_$4.log 2>&1
  echo "exit $?" >> gpurun_out/bench_$4.log
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$4.log").read().strip().splitlines()[-2])
    r=d["roofline"]
    print("$4", "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "fill_us", round(r["fill_kernel_avg_us"]), "order_us", round(r["order_kernel_avg_us"]))
    st=json.load(open("gpurun_out/stats_$4.json"))
    print({k:(round(v["mean"],1),round(v["max"],1)) for k,v in st.items() if isinstance(v,dict) and k in ("order_us","solver_iterations","solver_blocked","stager_iterations","p2_hist_quota_us","p2_keep_p3_p4_us")})
except Exception as e:
    print("$4 FAILED", e); print(open("gpurun_out/bench_$4.log").read()[-1500:])
PY
}
run 4 1 1 w4g1f1
run 4 2 1 w4g2f1
run 4 1 4 w4g1f4
run 4 2 4 w4g2f4
run 4 1 8 w4g1f8
run 4 2 8 w4g2f8
# the default line (with the CPU baseline)
timeout 600 python bench.py > gpurun_out/bench_default.log 2>&1; echo "exit $?" >> gpurun_out/bench_default.log
tail -2 gpurun_out/bench_default.log | cut -c1-600
# kernel trace of the default command
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -o trace -- python bench.py --no-cpu --check 0 > gpurun_out/prof_trace.log 2>&1; echo "trace exit $?"
# PMC passes (own runs, one counter each)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- python bench.py --no-cpu --check 0 --steps 2 --warmup 1 --in-flight 1 > gpurun_out/prof_fetch.log 2>&1; echo "fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o write -- python bench.py --no-cpu --check 0 --steps 2 --warmup 1 --in-flight 1 > gpurun_out/prof_write.log 2>&1; echo "write exit $?"
find gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write -type f | head -30
