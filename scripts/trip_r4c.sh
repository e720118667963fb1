#!/bin/bash
# relaxation form of P5, first time on the GPU: records against the ticket form of the same library and of round 3's library
O=gpurun_out/r4c; mkdir -p $O
P=kafka-assigner_amd/csrc/libkas_hip.so
run() { local name=$1; shift; timeout 100 "$@" > $O/$name.log 2>&1; echo "exit $?" >> $O/$name.log; grep -v "^   kas_" $O/$name.log | cut -c1-250; }
AB_INFLIGHT=8:20:3 run c3mix_relax tools/ab_harness c3mix 1000 5 $P
AB_FLAGS=65536 AB_INFLIGHT=8:20:3 run c3mix_ticket tools/ab_harness c3mix 1000 5 $P variants/libkas_hip_base.so
run c2_relax tools/ab_harness shape:10000:100:10:3 1 50 $P
AB_FLAGS=65536 run c2_ticket tools/ab_harness shape:10000:100:10:3 1 50 $P
run rf2 tools/ab_harness shape:50000:300:10:2 64 2 $P
AB_FLAGS=65536 run rf2_ticket tools/ab_harness shape:50000:300:10:2 64 2 $P
run n5000 tools/ab_harness shape:30000:5000:25:3 3 2 $P
AB_FLAGS=65536 run n5000_ticket tools/ab_harness shape:30000:5000:25:3 3 2 $P
run multi tools/ab_harness multi:20000:200:10 64 2 $P
AB_FLAGS=65536 run multi_ticket tools/ab_harness multi:20000:200:10 64 2 $P
