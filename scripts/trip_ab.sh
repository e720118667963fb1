#!/bin/bash
# scripts/trip_ab.sh NAME LIB...: tuning builds (variants/libkas_hip_LIB.so) through tools/ab_harness on one box, every slot its
# own tables: kernel durations alone, the in-flight rate at 8 x 40 and 8 x 20 steps, record checksums (must be equal).
O=gpurun_out/$1; shift; mkdir -p $O
export GPU_MAX_HW_QUEUES=16 AB_DISTINCT=1
LIBS=""; for v in "$@"; do LIBS="$LIBS variants/libkas_hip_$v.so"; done
for round in 1 2; do
  AB_INFLIGHT=8:40:3 timeout 300 tools/ab_harness c3mix 1000 3 $LIBS > $O/ab40_$round.log 2>&1; echo "exit $?" >> $O/ab40_$round.log
  grep -E "fill .* us|in flight|records|exit" $O/ab40_$round.log | cut -c1-200
done
AB_INFLIGHT=8:20:5 timeout 300 tools/ab_harness c3mix 1000 3 $LIBS > $O/ab20.log 2>&1; grep -E "in flight" $O/ab20.log | cut -c1-200
