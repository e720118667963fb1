#!/bin/bash
# one-off (round 3, last session): in-flight parity of the wide families with the new library against the old one, SQ instruction counters alone
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03g
mkdir -p $R/$O
export TMPDIR=/tmp
P=$R/kafka-assigner_amd/csrc/libkas_hip.so
OLD=$R/variants/libkas_hip_r3e.so
H=$R/tools/ab_harness
cd $R
AB_INFLIGHT=6:18:2 timeout 30 $H shape:100000:1000:40:5 96 2 $OLD $P > $O/ab_inflight_w5_96x6.log 2>&1; echo "exit $?" >> $O/ab_inflight_w5_96x6.log; grep -v "^   kas_" $O/ab_inflight_w5_96x6.log
AB_INFLIGHT=6:18:2 timeout 30 $H shape:100000:1000:40:4 96 2 $OLD $P > $O/ab_inflight_w4_96x6.log 2>&1; echo "exit $?" >> $O/ab_inflight_w4_96x6.log; grep -v "^   kas_" $O/ab_inflight_w4_96x6.log
AB_INFLIGHT=3:9:2 timeout 30 $H shape:1000000:5000:40:5 4 1 $OLD $P > $O/ab_inflight_c5_4x3.log 2>&1; echo "exit $?" >> $O/ab_inflight_c5_4x3.log; grep -v "^   kas_" $O/ab_inflight_c5_4x3.log
AB_INFLIGHT=8:24:2 AB_FLAGS=4096 timeout 30 $H c3mix 500 2 $OLD $P > $O/ab_inflight_c3mix_g1.log 2>&1; echo "exit $?" >> $O/ab_inflight_c3mix_g1.log; grep -v "^   kas_" $O/ab_inflight_c3mix_g1.log
cd /tmp
timeout 40 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/$O/prof_sq1_f1 -o sq1 -- $H c3mix 1000 2 $P > $R/$O/prof_sq1_f1.log 2>&1; echo "sq1 exit $?"
timeout 40 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $R/$O/prof_sq2_f1 -o sq2 -- $H c3mix 1000 2 $P > $R/$O/prof_sq2_f1.log 2>&1; echo "sq2 exit $?"
timeout 40 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $R/$O/prof_sq2_old_f1 -o sq2 -- $H c3mix 1000 2 $OLD > $R/$O/prof_sq2_old_f1.log 2>&1; echo "sq2 (old library) exit $?"
cd $R
