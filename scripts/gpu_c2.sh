#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
for lib in libkas_hip.so libkas_hip_g3.so; do
export KAS_HIP_LIB=$PWD/kafka-assigner_amd/csrc/$lib
echo "== $lib"
timeout 300 python bench.py --no-cpu --check 1 --scenarios 1 --partitions 10000 --brokers 100 --racks 10 --actions remove1 --in-flight 1 --steps 50 --warmup 5 --stats gpurun_out/stats_c2.json > gpurun_out/bench_c2x.log 2>&1
python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/bench_c2x.log").read().strip().splitlines() if l.startswith("{")][-1]
r=d["roofline"]
print("c2 ms/step", round(d["ms_per_step"],3), "fill_us", round(r["fill_kernel_avg_us"]), "order_us", round(r["order_kernel_avg_us"]))
st=json.load(open("gpurun_out/stats_c2.json"))
print({k:round(v["mean"],2) for k,v in st.items() if isinstance(v,dict) and k.startswith(("solver","p5","order"))})
PY
for f in 8 1; do
timeout 300 python bench.py --no-cpu --check 2 --steps 24 --warmup 4 --in-flight $f --stats gpurun_out/stats_x.json > gpurun_out/bench_x.log 2>&1
python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/bench_x.log").read().strip().splitlines() if l.startswith("{")][-1]
r=d["roofline"]
print("f$f value", round(d["value"]), "fill_us", round(r["fill_kernel_avg_us"]), "order_us", round(r["order_kernel_avg_us"]))
st=json.load(open("gpurun_out/stats_x.json"))
print({k:round(v["mean"],2) for k,v in st.items() if isinstance(v,dict) and k.startswith(("solver","p5","order"))})
PY
done
done
