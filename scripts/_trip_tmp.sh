#!/bin/bash
# one-off (round 3, last session): topic changes and rows narrower than the batch (the staging wave's rare paths) on the GPU, new library against the old one
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r03g
mkdir -p $R/$O
P=$R/kafka-assigner_amd/csrc/libkas_hip.so
OLD=$R/variants/libkas_hip_r3e.so
H=$R/tools/ab_harness
cd $R
AB_INFLIGHT=4:12:2 timeout 30 $H multi:100000:1000:20 200 2 $OLD $P > $O/ab_multi_3topics_200.log 2>&1; echo "exit $?" >> $O/ab_multi_3topics_200.log; grep -v "^   kas_" $O/ab_multi_3topics_200.log
timeout 20 $H multi:20000:300:10 64 2 $OLD $P > $O/ab_multi_3topics_64.log 2>&1; echo "exit $?" >> $O/ab_multi_3topics_64.log; grep -v "^   kas_" $O/ab_multi_3topics_64.log
AB_FLAGS=4096 timeout 20 $H multi:20000:300:10 64 2 $OLD $P > $O/ab_multi_3topics_64_g1.log 2>&1; echo "exit $?" >> $O/ab_multi_3topics_64_g1.log; grep -v "^   kas_" $O/ab_multi_3topics_64_g1.log
AB_FLAGS=4 timeout 20 $H multi:20000:300:10 64 2 $OLD $P > $O/ab_multi_3topics_64_unpacked.log 2>&1; echo "exit $?" >> $O/ab_multi_3topics_64_unpacked.log; grep -v "^   kas_" $O/ab_multi_3topics_64_unpacked.log
