#!/bin/bash
set -x
O=gpurun_out/r2l
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/host_path_small.py > $O/host_small.log 2>&1; cat $O/host_small.log | tail -6
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -10 $O/pytest_gpu.log
