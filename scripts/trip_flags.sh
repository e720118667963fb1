#!/bin/bash
# in-flight rate of the product library under plan flags (AB_FLAGS values given as arguments)
O=gpurun_out/r4k; mkdir -p $O
P=kafka-assigner_amd/csrc/libkas_hip.so
for f in "$@"; do
  AB_FLAGS=$f AB_INFLIGHT=8:24:3 timeout 60 tools/ab_harness c3mix 1000 3 $P > $O/flags_$f.log 2>&1
  echo "flags $f: $(grep -o 'fill *[0-9.]* us *order *[0-9.]* us' $O/flags_$f.log) | $(grep 'in flight' $O/flags_$f.log | cut -c1-100)"
done
