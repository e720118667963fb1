#!/bin/bash
export TMPDIR=/tmp
for g in 0 1 2; do
  for rep in 1 2; do
  timeout 300 python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 10000 --brokers 100 --racks 10 --actions remove1 --in-flight 1 --steps 100 --warmup 10 --groups $g --stats gpurun_out/c2_g$g.json | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']['in_flight_launch']; print('C2 groups=$g rep$rep ms_per_step', round(d['ms_per_step'],4), 'fill/order us', round(r['fill_kernel_us']), round(r['order_kernel_us']), d['roofline']['kernel'][:120])"
  done
done
python -c "
import json
for g in (0,1,2):
    d=json.load(open('gpurun_out/c2_g%d.json'%g)); print('C2 stats g',g,{k:round(v['mean']) for k,v in d.items() if isinstance(v,dict) and ('solver' in k or 'p5' in k or 'order' in k)})"
