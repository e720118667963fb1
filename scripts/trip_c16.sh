#!/bin/bash
# scripts/trip_c16.sh NAME [LIB]: one library, one box — the batch resident as int32 broker ids (kas_plan_create) and as 16-bit
# node-index cells (kas_plan_create16), tools/ab_harness, every slot its own tables.  MEASUREMENT TOOLING.
O=gpurun_out/$1; mkdir -p $O
LIB=${2:-kafka-assigner_amd/csrc/libkas_hip.so}
export GPU_MAX_HW_QUEUES=16 AB_DISTINCT=1
for round in 1 2; do
  for steps in 40 20; do
    AB_INFLIGHT=8:$steps:3 timeout 300 tools/ab_harness c3mix 1000 3 $LIB > $O/int32_${steps}_$round.log 2>&1
    echo "int32 ids    8 x $steps: $(grep -E "in flight" $O/int32_${steps}_$round.log | grep -o "[0-9.]*k scenarios/s" | tr '\n' ' ') $(grep -o "fill *[0-9.]* us *order *[0-9.]* us" $O/int32_${steps}_$round.log | head -1) $(grep -o "ok [0-9]*/[0-9]* moved [0-9]*" $O/int32_${steps}_$round.log | head -1)"
    AB_CELLS16=1 AB_INFLIGHT=8:$steps:3 timeout 300 tools/ab_harness c3mix 1000 3 $LIB > $O/c16_${steps}_$round.log 2>&1
    echo "16-bit cells 8 x $steps: $(grep -E "in flight" $O/c16_${steps}_$round.log | grep -o "[0-9.]*k scenarios/s" | tr '\n' ' ') $(grep -o "fill *[0-9.]* us *order *[0-9.]* us" $O/c16_${steps}_$round.log | head -1) $(grep -o "ok [0-9]*/[0-9]* moved [0-9]*" $O/c16_${steps}_$round.log | head -1)"
  done
done
