#!/bin/bash
# scripts/trip_slots.sh NAME [LIB]: batches in flight against the length of the timed region, one library, one box
# (tools/ab_harness, every slot its own tables): K slots x STEPS steps, five repeats each.  MEASUREMENT TOOLING.
O=gpurun_out/$1; mkdir -p $O
LIB=${2:-kafka-assigner_amd/csrc/libkas_hip.so}
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-24} AB_DISTINCT=1
for steps in 20 40; do
  for k in 8 10 12 16 20; do
    AB_INFLIGHT=$k:$steps:5 timeout 200 tools/ab_harness c3mix 1000 1 $LIB > $O/slots_${k}_$steps.log 2>&1
    echo "$k in flight x $steps steps: $(grep -E "in flight" $O/slots_${k}_$steps.log | grep -o "[0-9.]*k scenarios/s" | tr '\n' ' ')"
  done
done
