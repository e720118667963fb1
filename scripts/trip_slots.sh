#!/bin/bash
# how many batches in flight, now that every slot's stream has a hardware queue of its own (round 4 measured 6..12 equal — with two
# pairs of slots sharing a queue): tools/ab_harness, every slot its own tables, K x STEPS, under two queue limits
O=gpurun_out/${1:-slots}; mkdir -p $O
LIB=variants/libkas_hip_${2:-r5m}.so
export AB_DISTINCT=1
for q in 16 32; do
  for k in 6 8 10 12 16; do
    for st in 20 40; do
      GPU_MAX_HW_QUEUES=$q AB_INFLIGHT=$k:$st:4 timeout 120 tools/ab_harness c3mix 1000 1 $LIB > $O/ab_q${q}_k${k}_s$st.log 2>&1
      echo "queues $q slots $k steps $st: $(grep 'in flight' $O/ab_q${q}_k${k}_s$st.log | cut -c30-130)"
    done
  done
done
