#!/bin/bash
# parity tests + bench at several workgroup widths + kernel-trace profile
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
for nw in 4 2 8 1; do
  KAS_BENCH_WAVES=$nw timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu --check 4 --stats gpurun_out/stats_nw$nw.json > gpurun_out/bench_nw$nw.log 2>&1
  echo "bench nw=$nw exit $?" >> gpurun_out/bench_nw$nw.log
  tail -2 gpurun_out/bench_nw$nw.log | cut -c1-400
done
