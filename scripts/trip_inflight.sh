#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
P=kafka-assigner_amd/csrc/libkas_hip.so
for k in 4 6 8 10 12 16; do
  AB_INFLIGHT=$k:$((k*3)):3 timeout 60 tools/ab_harness c3mix 1000 2 $P > $O/inflight_$k.log 2>&1
  echo "in flight $k: $(grep 'in flight' $O/inflight_$k.log | cut -c1-120)"
done
