#!/bin/bash
# round-end evidence: parity tests, default bench line (with CPU baseline), smoke, kernel trace,
# PMC passes, plus the small BASELINE configs for the record
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 600 python bench.py --stats gpurun_out/stats_default.json > gpurun_out/bench_default.log 2>&1; echo "exit $?" >> gpurun_out/bench_default.log
tail -2 gpurun_out/bench_default.log | cut -c1-400
timeout 300 python bench.py --no-cpu --check 4 --in-flight 1 --stats gpurun_out/stats_f1.json > gpurun_out/bench_f1.log 2>&1
tail -1 gpurun_out/bench_f1.log | cut -c1-200
# BASELINE configs[1]: single scenario, 10k partitions x 100 brokers x 10 racks, RF 3 (latency)
timeout 300 python bench.py --no-cpu --check 1 --scenarios 1 --partitions 10000 --brokers 100 --racks 10 --actions remove1 --in-flight 1 --steps 50 --warmup 5 > gpurun_out/bench_c2.log 2>&1
tail -1 gpurun_out/bench_c2.log | cut -c1-300
# BASELINE configs[3] per-GPU share: 8000 scenarios in one batch (64k scenarios / 8 GPUs), add brokers
timeout 600 python bench.py --no-cpu --check 2 --scenarios 8000 --actions add_k --in-flight 2 --steps 4 --warmup 1 > gpurun_out/bench_c4.log 2>&1
tail -1 gpurun_out/bench_c4.log | cut -c1-300
# kernel trace of the default command
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -o trace -- python bench.py --no-cpu --check 0 > gpurun_out/prof_trace.log 2>&1; echo "trace exit $?"
# PMC passes (own runs, one counter each, one batch in flight)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o fetch -- python bench.py --no-cpu --check 0 --steps 2 --warmup 1 --in-flight 1 > gpurun_out/prof_fetch.log 2>&1; echo "fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o write -- python bench.py --no-cpu --check 0 --steps 2 --warmup 1 --in-flight 1 > gpurun_out/prof_write.log 2>&1; echo "write exit $?"
