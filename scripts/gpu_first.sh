#!/bin/bash
# first GPU contact: parity tests, short bench, kernel-trace profile
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 4 --warmup 1 --stats gpurun_out/stats.json > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log
tail -3 gpurun_out/bench.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python bench.py --steps 4 --warmup 1 --no-cpu --check 0 > gpurun_out/prof.log 2>&1
echo "prof exit $?" >> gpurun_out/prof.log
ls -R gpurun_out/prof | head -30
