#!/bin/bash
# scripts/trip_ab.sh NAME LIB [LIB ...]: configs[2] mix (1000 scenarios, alone and eight plans in flight) and configs[1]
# through the given builds of the library (tools/ab_harness): kernel durations, in-flight rate, record checksums
O=gpurun_out/$1; shift; mkdir -p $O
run() { local name=$1; shift; timeout 60 "$@" > $O/$name.log 2>&1; echo "exit $?" >> $O/$name.log; grep -v "^   kas_\|^exit 0\|generated in" $O/$name.log | cut -c1-250; }
AB_INFLIGHT=8:20:3 run c3mix tools/ab_harness c3mix 1000 5 "$@"
run c2 tools/ab_harness shape:10000:100:10:3 1 50 "$@"
