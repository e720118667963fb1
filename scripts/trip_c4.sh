#!/bin/bash
# configs[3]'s share on one GPU (8000 scenarios per batch, two batches in flight) through tools/ab_harness: libraries side by side
O=gpurun_out/${1:-c4}; shift; mkdir -p $O
export GPU_MAX_HW_QUEUES=16 AB_DISTINCT=1
LIBS=""; for v in "$@"; do LIBS="$LIBS variants/libkas_hip_$v.so"; done
AB_INFLIGHT=2:4:3 timeout 600 tools/ab_harness c3mix 8000 1 $LIBS > $O/ab_8000.log 2>&1; echo "exit $?" >> $O/ab_8000.log
grep -E "fill .* us|in flight|records|exit" $O/ab_8000.log | cut -c1-200
AB_INFLIGHT=3:6:3 timeout 600 tools/ab_harness c3mix 4000 1 $LIBS > $O/ab_4000.log 2>&1
grep -E "fill .* us|in flight" $O/ab_4000.log | cut -c1-200
