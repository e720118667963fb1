#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O; export TMPDIR=/tmp; R=$(pwd)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace_x64 -o trace -- python $R/bench.py --no-cpu --no-extras --repeats 1 --check 0 --scenarios 64 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 3 --warmup 1 > $R/$O/prof_trace_x64.log 2>&1; echo "trace x64 exit $?"
cd $R
grep "kas_" $O/prof_trace_x64/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
