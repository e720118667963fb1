#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --stats gpurun_out/stats_default.json > gpurun_out/bench_default.log 2>&1; echo "exit $?" >> gpurun_out/bench_default.log
tail -2 gpurun_out/bench_default.log | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -o trace -- python bench.py --no-cpu --check 0 > gpurun_out/prof_trace.log 2>&1; echo "trace exit $?"
