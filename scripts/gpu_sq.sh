#!/bin/bash
# SQ counter passes (one batch in flight): where the waves' cycles go
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
grep -c "" gpurun_out/counters_list.txt
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prof_sq1 -o sq1 -- python bench.py --no-cpu --check 0 --steps 2 --warmup 1 --in-flight 1 > gpurun_out/prof_sq1.log 2>&1; echo "sq1 exit $?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/prof_sq2 -o sq2 -- python bench.py --no-cpu --check 0 --steps 2 --warmup 1 --in-flight 1 > gpurun_out/prof_sq2.log 2>&1; echo "sq2 exit $?"
tail -3 gpurun_out/prof_sq1.log; tail -3 gpurun_out/prof_sq2.log
find gpurun_out/prof_sq1 gpurun_out/prof_sq2 -name "*.csv" | head
