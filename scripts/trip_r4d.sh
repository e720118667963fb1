#!/bin/bash
# relaxation form (pairs added in row-major order) against the ticket form of the same build; short timeouts
O=gpurun_out/r4d; mkdir -p $O
V=variants/libkas_hip_relax.so
run() { local name=$1; shift; timeout 30 "$@" > $O/$name.log 2>&1; echo "exit $?" >> $O/$name.log; grep -v "^   kas_" $O/$name.log | cut -c1-250; }
run c2_relax tools/ab_harness shape:10000:100:10:3 1 50 $V
AB_FLAGS=65536 run c2_ticket tools/ab_harness shape:10000:100:10:3 1 50 $V
run n5000 tools/ab_harness shape:30000:5000:25:3 3 2 $V
AB_FLAGS=65536 run n5000_ticket tools/ab_harness shape:30000:5000:25:3 3 2 $V
AB_INFLIGHT=8:20:3 run c3mix_relax tools/ab_harness c3mix 1000 5 $V
AB_FLAGS=65536 AB_INFLIGHT=8:20:3 run c3mix_ticket tools/ab_harness c3mix 1000 5 $V
run multi tools/ab_harness multi:20000:200:10 64 2 $V
AB_FLAGS=65536 run multi_ticket tools/ab_harness multi:20000:200:10 64 2 $V
