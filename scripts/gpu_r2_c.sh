#!/bin/bash
# round 2, trip C: fill-kernel register variants on configs[2]; wide-queue variants on configs[4]
set -x
O=gpurun_out/r2c
mkdir -p $O
export TMPDIR=/tmp
for v in mid5 mid4 mid5_ta3 mid5_nc mid4_nc mid6_nc; do
  KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout 200 python bench.py --no-cpu --check 8 --no-extras --steps 40 --stats $O/stats_$v.json > $O/bench_$v.log 2>&1; echo "exit $?" >> $O/bench_$v.log
  tail -2 $O/bench_$v.log | cut -c1-230
  KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout 200 python bench.py --no-cpu --check 0 --no-extras --steps 10 --in-flight 1 --stats $O/stats1_$v.json > $O/bench1_$v.log 2>&1
  grep -o '"in_flight_launch": {[^}]*' $O/bench1_$v.log | cut -c1-120
done
for v in w5_base w5_g1 w5_g1n1 w5_g1n1p3 w5_g1n1k8 w5_g1n1p3k8 w5_g2n2k8; do
  for act in c5 c5_norack; do
    KAS_HIP_LIB=$PWD/variants/libkas_hip_$v.so timeout 300 python bench.py --no-cpu --no-extras --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions $act --in-flight 1 --steps 6 --warmup 1 --stats $O/stats_${v}_$act.json > $O/bench_${v}_$act.log 2>&1; echo "exit $?" >> $O/bench_${v}_$act.log
    echo "$v $act $(grep -o '"in_flight_launch": {[^}]*' $O/bench_${v}_$act.log | cut -c1-110) $(tail -1 $O/bench_${v}_$act.log)"
  done
done
