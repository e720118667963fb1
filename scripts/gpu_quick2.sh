#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python bench.py --no-cpu --check 1 --scenarios 1 --partitions 10000 --brokers 100 --racks 10 --actions remove1 --in-flight 1 --steps 50 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c2', d['ms_per_step'], d['roofline']['fill_kernel_avg_us'], d['roofline']['order_kernel_avg_us'])"
for f in 8 1; do
timeout 300 python bench.py --no-cpu --check 4 --steps 24 --warmup 4 --in-flight $f --stats gpurun_out/stats_q.json 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f$f', d['value'], d['ms_per_step'], d['roofline']['fill_kernel_avg_us'], d['roofline']['order_kernel_avg_us'], d['config']['failed_scenarios_rank0'])"
python -c "
import json; st=json.load(open('gpurun_out/stats_q.json')); print({k:round(v['mean'],1) for k,v in st.items() if isinstance(v,dict) and k.startswith(('p2','p4','setup'))})"
done
