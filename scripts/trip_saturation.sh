#!/bin/bash
O=gpurun_out/${1:-r5c}; mkdir -p $O
export GPU_MAX_HW_QUEUES=16 AB_DISTINCT=1
for v in fo fonms; do
  for k in 8 4 2 1; do
    AB_INFLIGHT=$k:40:3 timeout 120 tools/ab_harness c3mix 1000 3 variants/libkas_hip_$v.so > $O/ab_${v}_$k.log 2>&1
    echo "$v x$k: $(grep -o 'fill .* us  order .* us' $O/ab_${v}_$k.log | head -1) | $(grep 'in flight' $O/ab_${v}_$k.log | cut -c1-120)"
  done
done
for st in 20 40 80; do
  AB_INFLIGHT=8:$st:3 timeout 120 tools/ab_harness c3mix 1000 3 variants/libkas_hip_r5p.so > $O/ab_r5p_steps$st.log 2>&1
  echo "r5p x8 steps $st: $(grep 'in flight' $O/ab_r5p_steps$st.log | cut -c1-120)"
done
for st in 20 40 80; do
  KAS_HIP_LIB=variants/libkas_hip_r5p.so timeout 300 python bench.py --no-cpu --check 0 --no-extras --repeats 3 --steps $st --warmup 5 > $O/bench_r5p_steps$st.log 2>&1
  echo "bench r5p steps $st: $(grep -o '"value": [0-9.]*' $O/bench_r5p_steps$st.log | head -1) $(grep -o '"values": \[[^]]*' $O/bench_r5p_steps$st.log | head -1)"
done
