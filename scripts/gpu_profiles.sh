#!/bin/bash
# scripts/gpu_profiles.sh NAME TAG: one GPU trip that leaves the SUMMARIES of profiles/ behind (gpurun copies at most 64 MiB
# back, and the raw kernel traces / counter files of twenty batches in flight are more): bench, traces, FETCH / WRITE and SQ
# counter passes (scripts/gpu_trip.sh), then collect_profiles.py / pipe_table.py / stream_timeline.py on the box, the results
# into gpurun_out/NAME/profiles/ and the raw rocprofv3 directories deleted.
NAME=$1; TAG=$2
O=gpurun_out/$NAME
bash scripts/gpu_trip.sh $NAME bench alone trace pmc sq
mkdir -p $O/profiles
cp -r profiles profiles.saved
python scripts/collect_profiles.py $O $TAG > $O/collect.log 2>&1; tail -3 $O/collect.log
python scripts/pipe_table.py $TAG $O/bench_driver_cmdline.log > $O/pipe_table.log 2>&1; tail -1 $O/pipe_table.log
python scripts/stream_timeline.py $O/prof_trace_default/trace_kernel_trace.csv profiles/${TAG}_stream_timeline.csv --skip-first 6 > $O/timeline.log 2>&1
for f in profiles/*; do cmp -s $f profiles.saved/$(basename $f) || cp $f $O/profiles/; done
rm -rf profiles; mv profiles.saved profiles
rm -rf $O/prof_trace* $O/prof_fetch $O/prof_write $O/prof_sq*_f*
du -sh $O
