#!/bin/bash
O=gpurun_out/${1:-bslots}; mkdir -p $O
for round in 1 2; do
for k in 8 10 12; do
  timeout 300 python bench.py --no-cpu --check 0 --no-extras --repeats 5 --steps 20 --warmup 5 --in-flight $k > $O/bench_k${k}_$round.log 2>&1
  echo "bench in-flight $k: $(grep -o '"value": [0-9.]*' $O/bench_k${k}_$round.log | head -1) $(grep -o '"values": \[[^]]*' $O/bench_k${k}_$round.log | head -1 | cut -c1-120)"
done
done
