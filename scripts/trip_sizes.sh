#!/bin/bash
# batch size x batches in flight, library by library (tools/ab_harness, every slot its own tables)
O=gpurun_out/${1:-sizes}; shift; mkdir -p $O
export GPU_MAX_HW_QUEUES=16 AB_DISTINCT=1
LIBS=""; for v in "$@"; do LIBS="$LIBS variants/libkas_hip_$v.so"; done
for cfg in 1000:8:40 1500:6:24 2000:4:16 3000:3:9; do
  IFS=: read S K ST <<< "$cfg"
  AB_INFLIGHT=$K:$ST:3 timeout 600 tools/ab_harness c3mix $S 1 $LIBS > $O/ab_$S.log 2>&1
  echo "== $S scenarios x $K in flight x $ST steps"; grep -E "fill .* us|in flight" $O/ab_$S.log | cut -c1-170
done
