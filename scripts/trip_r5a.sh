#!/bin/bash
# round 5, first GPU trip: the order kernel with its global accesses at one point per step (r5a; r5p = + s_setprio 3) against
# round 4's library — kernel durations alone and the in-flight rate (tools/ab_harness, same seeded batches), per-stream
# timelines of both (rocprofv3 --kernel-trace of the harness and of bench.py), bench.py's headline in alternation.
O=gpurun_out/${1:-r5a}; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GPU_MAX_HW_QUEUES=16 TMPDIR=/tmp
LIBS="variants/libkas_hip_r4.so variants/libkas_hip_r5a.so variants/libkas_hip_r5p.so"
AB_INFLIGHT=8:40:3 timeout 200 tools/ab_harness c3mix 1000 5 $LIBS > $O/ab_c3mix.log 2>&1; echo "exit $?" >> $O/ab_c3mix.log; cat $O/ab_c3mix.log
cd /tmp
for v in r4 r5a; do
  AB_INFLIGHT=8:40:2 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_ab_$v -o t -- $R/tools/ab_harness c3mix 1000 1 $R/variants/libkas_hip_$v.so > $R/$O/trace_ab_$v.log 2>&1
  f=$(find $R/$O/trace_ab_$v -name "*kernel_trace.csv" | head -1)
  python3 $R/scripts/stream_timeline.py $f $R/$O/timeline_ab_$v.csv --skip-first 2 > $R/$O/timeline_ab_$v.txt 2>&1; head -40 $R/$O/timeline_ab_$v.txt
done
cd $R
for round in 1 2; do
  for v in r4 r5a r5p; do
    KAS_HIP_LIB=variants/libkas_hip_$v.so timeout 300 python bench.py --no-cpu --check 0 --no-extras --repeats 3 --steps 20 --warmup 5 > $O/bench_${v}_$round.log 2>&1
    echo "bench $v $round: $(grep -o '"value": [0-9.]*' $O/bench_${v}_$round.log | head -1) $(grep -o '"values": \[[^]]*' $O/bench_${v}_$round.log | head -1) $(grep -o '"in_flight_launch": {[^}]*' $O/bench_${v}_$round.log | cut -c1-120)"
  done
done
cd /tmp
KAS_HIP_LIB=$R/variants/libkas_hip_r5a.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_bench_r5a -o t -- python $R/bench.py --no-cpu --check 0 --no-extras --repeats 2 --steps 20 --warmup 5 > $R/$O/trace_bench_r5a.log 2>&1
f=$(find $R/$O/trace_bench_r5a -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/stream_timeline.py $f $R/$O/timeline_bench_r5a.csv --skip-first 3 > $R/$O/timeline_bench_r5a.txt 2>&1; head -45 $R/$O/timeline_bench_r5a.txt
# keep the merge small: the raw traces stay on the box except the kas rows
for d in $R/$O/trace_*; do [ -d $d ] && for f in $(find $d -name "*kernel_trace.csv"); do grep -E "Kind|kas_" $f > $f.kas; rm -f $f; done; done
