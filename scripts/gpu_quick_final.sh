#!/bin/bash
# the driver's three round-end steps on the final tree: GPU tests, smoke, the bench command line
O=gpurun_out/${1:-final}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmdline.log 2>&1; echo "exit $?" >> $O/bench_driver_cmdline.log; tail -2 $O/bench_driver_cmdline.log | cut -c1-220
