#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --no-cpu --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions mixed --in-flight 1 --steps 3 --warmup 1 --stats gpurun_out/stats_c5.json > gpurun_out/bench_c5.log 2>&1
tail -1 gpurun_out/bench_c5.log | cut -c1-400
python - <<PY
import json
st=json.load(open("gpurun_out/stats_c5.json"))
print({k:(round(v["mean"],1)) for k,v in st.items() if isinstance(v,dict)})
PY
