#!/bin/bash
# timing experiment: 16-bit node-index cells in HBM (fill reads half the bytes, the order kernel gathers no broker ids) against
# the product, each on its own input form, same box.  MEASUREMENT TOOLING.
O=gpurun_out/$1; mkdir -p $O
export GPU_MAX_HW_QUEUES=16 AB_DISTINCT=1
for round in 1 2; do
  for steps in 40 20; do
    AB_INFLIGHT=8:$steps:3 timeout 300 tools/ab_harness c3mix 1000 3 variants/libkas_hip_cur.so > $O/cur_${steps}_$round.log 2>&1
    echo "int32 ids   8 x $steps: $(grep -E "in flight" $O/cur_${steps}_$round.log | grep -o "[0-9.]*k scenarios/s" | tr '\n' ' ') $(grep -o "fill *[0-9.]* us *order *[0-9.]* us" $O/cur_${steps}_$round.log | head -1) $(grep -o "ok [0-9]*/[0-9]* moved [0-9]*" $O/cur_${steps}_$round.log | head -1)"
    AB_CELLS16=1 AB_INFLIGHT=8:$steps:3 timeout 300 tools/ab_harness c3mix 1000 3 variants/libkas_hip_c16.so > $O/c16_${steps}_$round.log 2>&1
    echo "16-bit cells 8 x $steps: $(grep -E "in flight" $O/c16_${steps}_$round.log | grep -o "[0-9.]*k scenarios/s" | tr '\n' ' ') $(grep -o "fill *[0-9.]* us *order *[0-9.]* us" $O/c16_${steps}_$round.log | head -1) $(grep -o "ok [0-9]*/[0-9]* moved [0-9]*" $O/c16_${steps}_$round.log | head -1)"
  done
done
