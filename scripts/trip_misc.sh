#!/bin/bash
O=gpurun_out/${1:-misc}; mkdir -p $O
export GPU_MAX_HW_QUEUES=16 AB_DISTINCT=1
for round in 1 2; do
AB_INFLIGHT=8:40:3 timeout 300 tools/ab_harness c3mix 1000 1 variants/libkas_hip_cur.so variants/libkas_hip_curp0.so variants/libkas_hip_curp1.so > $O/prio_$round.log 2>&1; grep -E "in flight" $O/prio_$round.log | cut -c1-130
done
AB_FLAGS=0x40000 AB_INFLIGHT=8:40:3 timeout 300 tools/ab_harness c3mix 1000 1 variants/libkas_hip_cur.so > $O/dual.log 2>&1; echo "double tiles:"; grep -E "fill .* us|in flight" $O/dual.log | cut -c1-150
for k in 6 8 10 12; do
  AB_INFLIGHT=$k:20:5 timeout 300 tools/ab_harness c3mix 1000 1 variants/libkas_hip_cur.so > $O/slots_$k.log 2>&1; echo "slots $k x 20: $(grep 'in flight' $O/slots_$k.log | cut -c30-150)"
done
