#!/bin/bash
# quick A/B: KAS_HIP_LIB variants x batches in flight
mkdir -p gpurun_out
export TMPDIR=/tmp
run() {  # lib inflight
  KAS_HIP_LIB=$PWD/kafka-assigner_amd/csrc/$1 timeout 300 python bench.py --no-cpu --check 2 --steps 32 --warmup 8 --in-flight $2 --stats gpurun_out/stats_ab.json 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); st=json.load(open('gpurun_out/stats_ab.json'))
print('$1 f$2', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['fill_kernel_avg_us']), round(d['roofline']['order_kernel_avg_us']), 'steps', round(st['solver_iterations']['mean']), 'qsteps', round(st['p5_rounds_or_queue_steps']['mean']), 'qrows', round(st['solver_queue_rows']['mean']), 'order_us', round(st['order_us']['mean']))"
}
for l in libkas_hip.so libkas_hip_n1g2.so libkas_hip_n1g3.so libkas_hip_n3g3.so libkas_hip_n2g2.so libkas_hip_n2g4.so libkas_hip.so; do run $l 8; done
