#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O
V=variants/libkas_hip_prio.so
for pr in 0 1 0 1; do
  KAS_ORDER_STREAM_PRIORITY=$pr AB_INFLIGHT=8:24:3 timeout 60 tools/ab_harness c3mix 1000 2 $V > $O/prio_$pr.log 2>&1
  echo "order stream priority $pr: $(grep 'in flight' $O/prio_$pr.log | cut -c1-110) $(grep -o 'records [0-9a-f]*' $O/prio_$pr.log | head -1)"
done
