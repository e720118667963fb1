#!/bin/bash
# the in-flight consistency stress over several shapes / plan switches (scripts/stress_inflight.py)
O=gpurun_out/${1:-stress}; mkdir -p $O
export TMPDIR=/tmp
i=0
run() { i=$((i+1)); timeout 300 python scripts/stress_inflight.py "$@" > $O/s_$i.log 2>&1; echo "[$*] exit $? $(grep 'solves of' $O/s_$i.log | cut -c1-150) | $(grep '^plan:' $O/s_$i.log | cut -c1-200)"; }
run 500
run 150 --groups 1
run 150 --plan-flags 4
run 150 --plan-flags 8
run 60 --scenarios 4000 --actions add50 --in-flight 3
run 100 --scenarios 4096 --partitions 10000 --brokers 100 --racks 10 --in-flight 8
run 100 --scenarios 96 --partitions 40000 --brokers 600 --racks 40 --rf 5 --actions add_k,mixed --in-flight 6
run 100 --scenarios 96 --partitions 40000 --brokers 600 --racks 40 --rf 4 --actions add_k,mixed,remove_k --in-flight 6
run 40 --scenarios 16 --partitions 400000 --brokers 2000 --racks 40 --rf 5 --actions add_k,mixed --in-flight 3
