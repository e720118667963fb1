#!/bin/bash
# round 4, first GPU trip: the issue-rate probe (what a SIMD issues per cycle: the denominators of the per-pipe table),
# then the prepared patches of round 3 one by one through tools/ab_harness (seconds each).
O=gpurun_out/r4a; mkdir -p $O
P=kafka-assigner_amd/csrc/libkas_hip.so
timeout 120 tools/issue_probe 300 > $O/issue_probe.log 2>&1; echo "issue_probe exit $?"; grep -c . $O/issue_probe.log
run() { local name=$1; shift; timeout 100 "$@" > $O/$name.log 2>&1; echo "exit $?" >> $O/$name.log; grep -v "^   kas_" $O/$name.log | cut -c1-230; }
run c5_wide tools/ab_harness c5 1 3 $P variants/libkas_hip_base5.so variants/libkas_hip_lds.so variants/libkas_hip_wflat.so variants/libkas_hip_wside.so
run c5norack_wide tools/ab_harness c5norack 1 2 $P variants/libkas_hip_lds.so
run c2_claim tools/ab_harness shape:10000:100:10:3 1 50 $P variants/libkas_hip_claim.so
AB_INFLIGHT=8:20:3 run c3mix_claim tools/ab_harness c3mix 1000 5 $P variants/libkas_hip_claim.so
