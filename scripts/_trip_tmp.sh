#!/bin/bash
O=gpurun_out/r4k; mkdir -p $O
for v in c5p4old c5slim; do
  KAS_HIP_LIB=variants/libkas_hip_$v.so timeout 200 python bench.py --no-cpu --no-extras --repeats 1 --check 2 --scenarios 64 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 3 --warmup 1 > $O/x64_$v.log 2>&1
  echo "$v x64: rc=$? ms_per_step $(grep -o '"ms_per_step": [0-9.]*' $O/x64_$v.log | head -1) $(grep -o '"in_flight_launch": {[^}]*' $O/x64_$v.log | cut -c1-110)"
  KAS_HIP_LIB=variants/libkas_hip_$v.so timeout 100 python bench.py --no-cpu --no-extras --repeats 1 --check 2 --scenarios 8 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 5 --warmup 1 > $O/x8_$v.log 2>&1
  echo "$v x8: ms_per_step $(grep -o '"ms_per_step": [0-9.]*' $O/x8_$v.log | head -1) $(grep -o '"in_flight_launch": {[^}]*' $O/x8_$v.log | cut -c1-110)"
done
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
KAS_HIP_LIB=$R/variants/libkas_hip_c5slim.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_x64 -o trace -- python $R/bench.py --no-cpu --no-extras --repeats 1 --check 0 --scenarios 64 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 3 --warmup 1 > $R/$O/prof_x64.log 2>&1
cd $R; grep "kas_" $O/prof_x64/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
