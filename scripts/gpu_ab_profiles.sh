#!/bin/bash
# scripts/gpu_ab_profiles.sh OLD.so — evidence for a kernel change from the short end of a GPU budget, with tools/ab_harness
# (no torch import) as the workload: A/B of the in-tree library against OLD.so (same seeded batches: kernel durations,
# in-flight rate, record checksums), then rocprofv3 on the in-tree library: --pmc FETCH_SIZE / WRITE_SIZE passes and
# --kernel-trace --stats (one batch alone, configs[4], eight plans in flight) on BASELINE configs[2]'s kernels
# (c3mix 1000: the bench's plan string).  scripts/collect_profiles.py DIR TAG turns the CSVs into profiles/.
#   gpurun --timeout 110 -- 'bash scripts/gpu_ab_profiles.sh variants/libkas_hip_r3e.so'
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/${2:-r03g}
mkdir -p $R/$O
export TMPDIR=/tmp
P=$R/kafka-assigner_amd/csrc/libkas_hip.so
OLD=$R/$1
H=$R/tools/ab_harness
cd $R
AB_INFLIGHT=8:20:3 timeout 40 $H c3mix 1000 5 $OLD $P > $O/ab_c3mix_1000.log 2>&1; echo "exit $?" >> $O/ab_c3mix_1000.log; grep -v "^   kas_" $O/ab_c3mix_1000.log
timeout 20 $H c5 1 3 $OLD $P > $O/ab_c5.log 2>&1; echo "exit $?" >> $O/ab_c5.log; grep -v "^   kas_" $O/ab_c5.log
timeout 20 $H c5norack 1 3 $OLD $P > $O/ab_c5norack.log 2>&1; echo "exit $?" >> $O/ab_c5norack.log; grep -v "^   kas_" $O/ab_c5norack.log
cd /tmp
timeout 40 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_fetch -o fetch -- $H c3mix 1000 2 $P > $R/$O/prof_fetch.log 2>&1; echo "fetch exit $?"
timeout 40 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_write -o write -- $H c3mix 1000 2 $P > $R/$O/prof_write.log 2>&1; echo "write exit $?"
timeout 40 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace_one_batch_in_flight -o trace -- $H c3mix 1000 10 $P > $R/$O/prof_trace_f1.log 2>&1; echo "trace (alone) exit $?"
timeout 40 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace_config5 -o trace -- $H c5 1 5 $P > $R/$O/prof_trace_c5.log 2>&1; echo "trace (configs[4]) exit $?"
AB_INFLIGHT=8:20:2 timeout 40 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace_default -o trace -- $H c3mix 1000 1 $P > $R/$O/prof_trace_default.log 2>&1; echo "trace (in flight) exit $?"
cd $R
ls $O
