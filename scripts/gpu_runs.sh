#!/bin/bash
# A/B of the run path in the ticket-form solver: GPU parity suite, then bench new vs previous library
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/runs_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/runs_pytest.log
tail -3 gpurun_out/runs_pytest.log
run() {  # groups inflight tag
  timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu --check 4 --groups $1 --in-flight $2 --stats gpurun_out/stats_$3.json > gpurun_out/bench_$3.log 2>&1
  echo "exit $?" >> gpurun_out/bench_$3.log
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open("gpurun_out/bench_$3.log").read().strip().splitlines() if l.startswith("{")][-1]
    r=d["roofline"]
    print("$3", "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "fill_us", round(r["fill_kernel_avg_us"]), "order_us", round(r["order_kernel_avg_us"]))
    st=json.load(open("gpurun_out/stats_$3.json"))
    print({k:(round(v["mean"],1),round(v["max"],1)) for k,v in st.items() if isinstance(v,dict) and k in ("order_us","solver_iterations","solver_run_rounds","solver_run_rows","solver_blocked","stager_iterations","stager_idle")})
except Exception as e:
    print("$3 FAILED", e); print(open("gpurun_out/bench_$3.log").read()[-1500:])
PY
}
run 2 8 run_g2f8
run 2 1 run_g2f1
run 1 8 run_g1f8
run 1 1 run_g1f1
export KAS_HIP_LIB=$PWD/kafka-assigner_amd/csrc/libkas_hip_prev.so
run 2 8 prev_g2f8
unset KAS_HIP_LIB
run 2 8 run_g2f8b
