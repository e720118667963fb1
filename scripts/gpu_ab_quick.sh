#!/bin/bash
# scripts/gpu_ab_quick.sh — the short end of a GPU budget (no torch import, seconds per step): tools/ab_harness runs the
# same seeded batch through the product library and a tuning build (scripts/build_variant.sh) and prints the fill / order
# kernel durations of each and a checksum of the result records, which must be equal (and equal to the emulator's:
# AB_EMU=tests/emu/libkas_emu.so tools/ab_harness ...).  gpurun --timeout 150 -- 'bash scripts/gpu_ab_quick.sh'
O=gpurun_out/abq
mkdir -p $O
P=kafka-assigner_amd/csrc/libkas_hip.so
timeout 60 tools/ab_harness c5 1 3 $P variants/libkas_hip_widepick.so > $O/c5.log 2>&1; echo "exit $?" >> $O/c5.log; cat $O/c5.log
timeout 90 tools/ab_harness c3 1000 10 $P variants/libkas_hip_stager.so > $O/c3.log 2>&1; echo "exit $?" >> $O/c3.log; cat $O/c3.log
