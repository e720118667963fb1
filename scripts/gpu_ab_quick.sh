#!/bin/bash
# scripts/gpu_ab_quick.sh [VARIANT]: the short end of a GPU budget (no torch import, seconds per step): tools/ab_harness runs
# the same seeded batches through the product library and a build of a patched tree (default variants/libkas_hip_r3f.so:
# every instance, experiments/stager_straight_line_read_ahead.patch + wide_pick_keys_name_their_position.patch) and prints
# the fill / order kernel durations of each, the in-flight rate, and a checksum of the result records, which must be equal
# (and equal to the emulator's: AB_EMU=tests/emu/libkas_emu.so tools/ab_harness ...).
#   gpurun --timeout 150 -- 'bash scripts/gpu_ab_quick.sh'
O=gpurun_out/abq
mkdir -p $O
P=kafka-assigner_amd/csrc/libkas_hip.so
V=${1:-variants/libkas_hip_r3f.so}
run() { local name=$1; shift; timeout 80 "$@" $P $V > $O/$name.log 2>&1; echo "exit $?" >> $O/$name.log; cat $O/$name.log; }
AB_INFLIGHT=8:20:3 run c3mix_1000 tools/ab_harness c3mix 1000 5
run c5 tools/ab_harness c5 1 3
run c5norack tools/ab_harness c5norack 1 2
run w4 tools/ab_harness shape:200000:1000:40:4 4 2
run rf2 tools/ab_harness shape:50000:300:10:2 64 2
AB_FLAGS=4096 run c3mix_g1 tools/ab_harness c3mix 200 2
AB_FLAGS=4 run c3mix_unpacked tools/ab_harness c3mix 200 2
run n5000 tools/ab_harness shape:30000:5000:25:3 3 2
