#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
run() {  # lib groups inflight tag
  KAS_HIP_LIB=$PWD/kafka-assigner_amd/csrc/$1 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu --check 2 --groups $2 --in-flight $3 --stats gpurun_out/stats_$4.json > gpurun_out/bench_$4.log 2>&1
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open("gpurun_out/bench_$4.log").read().strip().splitlines() if l.startswith("{")][-1]
    r=d["roofline"]
    print("$4", "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "fill_us", round(r["fill_kernel_avg_us"]), "order_us", round(r["order_kernel_avg_us"]))
    st=json.load(open("gpurun_out/stats_$4.json"))
    print({k:round(v["mean"],2) for k,v in st.items() if isinstance(v,dict)})
except Exception as e:
    print("$4 FAILED", e); print(open("gpurun_out/bench_$4.log").read()[-1500:])
PY
}
run libkas_hip_dbg.so 2 1 dbg_f1
run libkas_hip.so 2 1 k4_f1
run libkas_hip_k8.so 2 1 k8_f1
run libkas_hip.so 2 8 k4_f8
run libkas_hip_k8.so 2 8 k8_f8
run libkas_hip.so 1 8 k4_g1f8
run libkas_hip_k8.so 1 8 k8_g1f8
