#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # groups inflight tag extra
  timeout 300 python bench.py --steps 16 --warmup 4 --no-cpu --check 4 --groups $1 --in-flight $2 --stats gpurun_out/stats_$3.json $4 > gpurun_out/bench_$3.log 2>&1
  echo "exit $?" >> gpurun_out/bench_$3.log
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open("gpurun_out/bench_$3.log").read().strip().splitlines() if l.startswith("{")][-1]
    r=d["roofline"]
    print("$3", "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "fill_us", round(r["fill_kernel_avg_us"]), "order_us", round(r["order_kernel_avg_us"]))
    st=json.load(open("gpurun_out/stats_$3.json"))
    print({k:(round(v["mean"],1),round(v["max"],1)) for k,v in st.items() if isinstance(v,dict) and k in ("order_us","solver_iterations","solver_run_rounds","solver_run_rows","solver_blocked","stager_iterations","stager_idle","p5_rounds","solver_rows_in_hand","p4_windows")})
except Exception as e:
    print("$3 FAILED", e); print(open("gpurun_out/bench_$3.log").read()[-1500:])
PY
}
run 2 8 i_g2f8
run 2 1 i_g2f1
run 2 1 i_addk "--actions add_k"
run 2 1 i_rem1 "--actions remove1"
run 2 1 i_remk "--actions remove_k"
run 1 1 i_addk_g1 "--actions add_k"
