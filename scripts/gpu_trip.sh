#!/bin/bash
# One GPU trip (gpurun -- 'bash scripts/gpu_trip.sh NAME STEP...'): every step writes its log under
# gpurun_out/NAME/ and prints one summary line, so that a trip's outcome can be read from the tail.
#   tests      pytest -m gpu (the driver's round-end suite)
#   smoke      __graft_entry__.smoke()
#   bench      bench.py at the driver's command line (--gpus 1 --steps 20 --warmup 5)
#   benchq     bench.py without the CPU / end-to-end / other-config legs (quick headline figure + counters)
#   alone      one batch in flight (per-kernel durations free of other launches)
#   configs    BASELINE configs[1], configs[3]'s share, configs[4] (x1 rack map on / off, x8, x64)
#   stress     scripts/stress_inflight.py --suite 1000
#   trace      rocprofv3 --kernel-trace --stats: one batch alone, default, configs[4]
#   pmc        rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (one batch in flight)
#   sq         rocprofv3 --pmc SQ_* passes, one batch alone and eight in flight
#   ab:LIB     benchq with KAS_HIP_LIB=variants/libkas_hip_LIB.so (tuning builds, scripts/build_variant.sh)
#   abh:LIB    tools/ab_harness (no Python: seconds): product against variants/libkas_hip_LIB.so on seeded batches — kernel
#              durations, in-flight rate, record checksums (scripts/gpu_ab_quick.sh; LIB must hold every kernel the batches
#              launch: build it with -- -DKAS_MINIMAL_INSTANCES=0)
#   c5:LIB     configs[4] x1 with that tuning build (LIB = - for the product library)
#   random     scripts/stress_gpu.py 150: random shapes against the oracle, every plan variant
#   big        shapes beyond round 2's limits: 1.1M x 5k x RF 5 (checked wide form / round form), 1M x 5k x RF 3 (ticket / round form)
set -u
NAME=$1; shift
O=gpurun_out/$NAME
mkdir -p $O
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
val() { grep -o "\"$1\": [0-9.]*" "$2" | head -1 | cut -d' ' -f2; }
c5() {  # $1 = log stem, extra bench flags follow
  local stem=$1; shift
  timeout 300 python bench.py --no-cpu --no-extras --repeats 1 --check 1 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --in-flight 1 --steps 6 --warmup 1 "$@" > $O/$stem.log 2>&1
  echo "$stem: ms_per_step $(val ms_per_step $O/$stem.log) $(grep -o '"in_flight_launch": {[^}]*' $O/$stem.log | cut -c1-110)"
}
for step in "$@"; do
  case $step in
    tests)
      timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
      tail -12 $O/pytest_gpu.log | cut -c1-220 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -2 $O/smoke.log | cut -c1-200 ;;
    bench)
      timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmdline.log 2>&1; echo "exit $?" >> $O/bench_driver_cmdline.log
      echo "bench: value $(val value $O/bench_driver_cmdline.log) $(grep -o '"repeats": {[^}]*' $O/bench_driver_cmdline.log | cut -c1-260)"; tail -1 $O/bench_driver_cmdline.log | cut -c1-120
      grep -v '^{' $O/bench_driver_cmdline.log | tail -5 | cut -c1-300 ;;
    benchq)
      timeout 600 python bench.py --no-cpu --check 0 --no-extras --stats $O/stats_default.json > $O/bench_quick.log 2>&1; echo "exit $?" >> $O/bench_quick.log
      echo "benchq: value $(val value $O/bench_quick.log) $(grep -o '"values": \[[^]]*' $O/bench_quick.log | cut -c1-200)" ;;
    alone)
      timeout 300 python bench.py --no-cpu --check 0 --no-extras --repeats 1 --steps 10 --in-flight 1 --stats $O/stats_one_batch_in_flight.json > $O/bench_one_batch_in_flight.log 2>&1
      echo "alone: ms_per_step $(val ms_per_step $O/bench_one_batch_in_flight.log) $(grep -o '"in_flight_launch": {[^}]*' $O/bench_one_batch_in_flight.log | cut -c1-110)" ;;
    configs)
      timeout 300 python bench.py --no-cpu --no-extras --repeats 1 --check 1 --scenarios 1 --partitions 10000 --brokers 100 --racks 10 --actions remove1 --in-flight 1 --steps 50 --warmup 5 > $O/bench_config2_single_scenario.log 2>&1
      echo "configs[1]: ms_per_step $(val ms_per_step $O/bench_config2_single_scenario.log)"
      timeout 900 python bench.py --no-cpu --no-extras --repeats 3 --check 64 --scenarios 8000 --actions add50 --in-flight 2 --steps 4 --warmup 1 > $O/bench_config4_8000_scenarios_add50.log 2>&1
      echo "configs[3] share: value $(val value $O/bench_config4_8000_scenarios_add50.log)"
      c5 bench_config5_c5 --actions c5 --stats $O/stats_config5_c5.json
      c5 bench_config5_c5_norack --actions c5_norack --stats $O/stats_config5_c5_norack.json
      timeout 300 python bench.py --no-cpu --no-extras --repeats 1 --check 2 --scenarios 8 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 5 --warmup 1 > $O/bench_config5_x8.log 2>&1
      echo "configs[4] x8: ms_per_step $(val ms_per_step $O/bench_config5_x8.log)"
      timeout 600 python bench.py --no-cpu --no-extras --repeats 1 --check 2 --scenarios 64 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 3 --warmup 1 > $O/bench_config5_x64.log 2>&1
      echo "configs[4] x64: ms_per_step $(val ms_per_step $O/bench_config5_x64.log) $(grep -o '"in_flight_launch": {[^}]*' $O/bench_config5_x64.log | cut -c1-110)"
      # beyond the 10-bit bound of the wide form's count fields (1,079 rows per broker): wide form with its check, and the round form it used to drop to
      c5 bench_1100k_5k_rf5_wide_checked --actions c5 --partitions 1100000
      c5 bench_1100k_5k_rf5_round_form --actions c5 --partitions 1100000 --plan-flags 2 --steps 1 --warmup 1 ;;
    stress)
      timeout 900 python scripts/stress_inflight.py --suite 1000 > $O/stress_inflight.log 2>&1; echo "stress exit $?" >> $O/stress_inflight.log; grep -v "plan:" $O/stress_inflight.log | cut -c1-220 ;;
    trace)
      cd /tmp
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace_one_batch_in_flight -o trace -- python $R/bench.py --no-cpu --check 0 --no-extras --repeats 1 --in-flight 1 --steps 20 > $R/$O/prof_trace_f1.log 2>&1; echo "trace (alone) exit $?"
      timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace_default -o trace -- python $R/bench.py --no-cpu --check 0 --no-extras --repeats 2 > $R/$O/prof_trace_default.log 2>&1; echo "trace (default) exit $?"
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace_config5 -o trace -- python $R/bench.py --no-cpu --no-extras --repeats 1 --check 0 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 6 --warmup 1 > $R/$O/prof_trace_c5.log 2>&1; echo "trace (configs[4]) exit $?"
      cd $R ;;
    pmc)
      cd /tmp
      timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_fetch -o fetch -- python $R/bench.py --no-cpu --check 0 --no-extras --repeats 1 --steps 2 --warmup 1 --in-flight 1 > $R/$O/prof_fetch.log 2>&1; echo "fetch exit $?"
      timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_write -o write -- python $R/bench.py --no-cpu --check 0 --no-extras --repeats 1 --steps 2 --warmup 1 --in-flight 1 > $R/$O/prof_write.log 2>&1; echo "write exit $?"
      cd $R ;;
    sq)
      cd /tmp
      for mode in 1 ${KAS_BENCH_SLOTS:-12}; do
        timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/$O/prof_sq1_f$mode -o sq1 -- python $R/bench.py --no-cpu --check 0 --no-extras --repeats 1 --steps $((2 * mode)) --warmup 1 --in-flight $mode > $R/$O/prof_sq1_f$mode.log 2>&1; echo "sq1 f$mode exit $?"
        timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $R/$O/prof_sq2_f$mode -o sq2 -- python $R/bench.py --no-cpu --check 0 --no-extras --repeats 1 --steps $((2 * mode)) --warmup 1 --in-flight $mode > $R/$O/prof_sq2_f$mode.log 2>&1; echo "sq2 f$mode exit $?"
        # per-pipe: cycles a pipe spends on instructions (quad-cycle units, summed over waves), round 4
        timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $R/$O/prof_sq3_f$mode -o sq3 -- python $R/bench.py --no-cpu --check 0 --no-extras --repeats 1 --steps $((2 * mode)) --warmup 1 --in-flight $mode > $R/$O/prof_sq3_f$mode.log 2>&1; echo "sq3 f$mode exit $?"
        timeout 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$O/prof_sq4_f$mode -o sq4 -- python $R/bench.py --no-cpu --check 0 --no-extras --repeats 1 --steps $((2 * mode)) --warmup 1 --in-flight $mode > $R/$O/prof_sq4_f$mode.log 2>&1; echo "sq4 f$mode exit $?"
      done
      cd $R ;;
    sqc5)   # SQ counter passes of configs[4] x1 (1M x 5k x RF 5, one scenario): where kas_order_wide_kernel<5>'s time goes
      cd /tmp
      C5="--no-cpu --no-extras --repeats 1 --check 0 --scenarios 1 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 3 --warmup 1"
      timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/$O/prof_c5sq1 -o sq1 -- python $R/bench.py $C5 > $R/$O/prof_c5sq1.log 2>&1; echo "c5 sq1 exit $?"
      timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $R/$O/prof_c5sq2 -o sq2 -- python $R/bench.py $C5 > $R/$O/prof_c5sq2.log 2>&1; echo "c5 sq2 exit $?"
      timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $R/$O/prof_c5sq3 -o sq3 -- python $R/bench.py $C5 > $R/$O/prof_c5sq3.log 2>&1; echo "c5 sq3 exit $?"
      timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/$O/prof_c5sq4 -o sq4 -- python $R/bench.py $C5 > $R/$O/prof_c5sq4.log 2>&1; echo "c5 sq4 exit $?"
      timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace_config5 -o trace -- python $R/bench.py $C5 > $R/$O/prof_trace_c5.log 2>&1; echo "c5 trace exit $?"
      cd $R ;;
    ab32:*)   # benchq with that tuning build and extra plan flags: ab32:LIB:FLAGS (e.g. ab32:base:64 = no index rows)
      rest=${step#ab32:}; lib=${rest%%:*}; pf=${rest#*:}; [ "$pf" == "$rest" ] && pf=0
      KAS_HIP_LIB=variants/libkas_hip_$lib.so timeout 400 python bench.py --no-cpu --check 0 --no-extras --repeats 3 --plan-flags $pf > $O/bench_v_${lib}_pf$pf.log 2>&1
      echo "ab32 $lib flags $pf: value $(val value $O/bench_v_${lib}_pf$pf.log) $(grep -o '"values": \[[^]]*' $O/bench_v_${lib}_pf$pf.log | cut -c1-120)"
      grep -v '^{' $O/bench_v_${lib}_pf$pf.log | tail -2 | cut -c1-200 ;;
    ab1f:*)   # one batch alone with that tuning build and plan flags: ab1f:LIB:FLAGS
      rest=${step#ab1f:}; lib=${rest%%:*}; pf=${rest#*:}; [ "$pf" == "$rest" ] && pf=0
      KAS_HIP_LIB=variants/libkas_hip_$lib.so timeout 400 python bench.py --no-cpu --check 0 --no-extras --repeats 1 --steps 10 --in-flight 1 --plan-flags $pf > $O/bench_v1_${lib}_pf$pf.log 2>&1
      echo "ab1f $lib flags $pf: ms_per_step $(val ms_per_step $O/bench_v1_${lib}_pf$pf.log) $(grep -o '"in_flight_launch": {[^}]*' $O/bench_v1_${lib}_pf$pf.log | cut -c1-110)" ;;
    abh:*)
      bash scripts/gpu_ab_quick.sh variants/libkas_hip_${step#abh:}.so | grep -v "^   kas_" ;;
    ab:*)
      lib=${step#ab:}
      KAS_HIP_LIB=variants/libkas_hip_$lib.so timeout 400 python bench.py --no-cpu --check 0 --no-extras --repeats 3 --stats $O/stats_v_$lib.json > $O/bench_v_$lib.log 2>&1
      echo "ab $lib: value $(val value $O/bench_v_$lib.log) $(grep -o '"values": \[[^]]*' $O/bench_v_$lib.log | cut -c1-120) $(grep -o '"one_batch_alone": {[^}]*' $O/bench_v_$lib.log | cut -c1-140)"
      python - $O/stats_v_$lib.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("   stats:", " ".join("%s %.0f" % (k, d[k]["mean"]) for k in ("solver_iterations", "solver_blocked", "stager_iterations", "stager_idle", "p5_rounds_or_queue_steps", "order_us", "solver_rows_in_hand", "p2_ranked_tiles_wave0", "p4_windows", "p4_steps") if k in d))
except Exception as e:
    print("   no stats:", e)
PY
      grep -v '^{' $O/bench_v_$lib.log | tail -2 | cut -c1-200 ;;
    ab1:*)
      lib=${step#ab1:}
      KAS_HIP_LIB=variants/libkas_hip_$lib.so timeout 400 python bench.py --no-cpu --check 0 --no-extras --repeats 1 --steps 10 --in-flight 1 --stats $O/stats_v1_$lib.json > $O/bench_v1_$lib.log 2>&1
      echo "ab1 $lib: ms_per_step $(val ms_per_step $O/bench_v1_$lib.log) $(grep -o '"in_flight_launch": {[^}]*' $O/bench_v1_$lib.log | cut -c1-110)"
      python - $O/stats_v1_$lib.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("   stats:", " ".join("%s %.0f" % (k, d[k]["mean"]) for k in ("solver_iterations", "solver_blocked", "stager_iterations", "stager_idle", "p5_rounds_or_queue_steps", "order_us", "solver_rows_in_hand", "p2_ranked_tiles_wave0", "p4_windows", "p4_steps") if k in d))
PY
      ;;
    pcie)     # the host link: copies of the host path's sizes and of 256 MB, pageable / pinned, one and both directions
      timeout 120 tools/pcie_probe 36 8 > $O/pcie_36MB.log 2>&1; timeout 120 tools/pcie_probe 256 4 > $O/pcie_256MB.log 2>&1; tail -n 30 $O/pcie_256MB.log | cut -c1-130 ;;
    probe)    # what a SIMD issues per cycle (tools/issue_probe)
      timeout 200 tools/issue_probe 300 > $O/issue_probe.log 2>&1; echo "issue_probe exit $?" ;;
    random)   # randomised GPU-vs-oracle stress, 150 s of batches
      timeout 400 python scripts/stress_gpu.py 150 ${STRESS_SEED:-2026} > $O/stress_random.log 2>&1; echo "random stress exit $?" >> $O/stress_random.log; tail -2 $O/stress_random.log ;;
    big)
      c5 bench_1100k_5k_rf5_wide_checked --actions c5 --partitions 1100000
      c5 bench_1100k_5k_rf5_round_form --actions c5 --partitions 1100000 --plan-flags 2 --steps 1 --warmup 1
      # lists 3 wide at 5,000 brokers: one group of the ticket form (round 2: the round form from 4,680 brokers on)
      c5 bench_1m_5k_rf3_ticket_form --actions c5 --rf 3
      c5 bench_1m_5k_rf3_round_form --actions c5 --rf 3 --plan-flags 2 --steps 1 --warmup 1 ;;
    c5f:*)    # configs[4] x1 (and the batch of 8) with that tuning build and plan flags: c5f:LIB:FLAGS (131072 = the relaxation form for wide lists)
      rest=${step#c5f:}; lib=${rest%%:*}; pf=${rest#*:}; [ "$pf" == "$rest" ] && pf=0
      KAS_HIP_LIB=variants/libkas_hip_$lib.so c5 bench_c5_v_${lib}_pf$pf --actions c5 --plan-flags $pf --stats $O/stats_c5_${lib}_pf$pf.json
      KAS_HIP_LIB=variants/libkas_hip_$lib.so c5 bench_c5norack_v_${lib}_pf$pf --actions c5_norack --plan-flags $pf
      KAS_HIP_LIB=variants/libkas_hip_$lib.so timeout 300 python bench.py --no-cpu --no-extras --repeats 1 --check 2 --scenarios 64 --partitions 1000000 --brokers 5000 --racks 40 --rf 5 --actions c5 --in-flight 1 --steps 3 --warmup 1 --plan-flags $pf > $O/bench_c5x64_v_${lib}_pf$pf.log 2>&1
      echo "configs[4] x64 $lib flags $pf: ms_per_step $(val ms_per_step $O/bench_c5x64_v_${lib}_pf$pf.log) $(grep -o '"in_flight_launch": {[^}]*' $O/bench_c5x64_v_${lib}_pf$pf.log | cut -c1-110)"
      python - $O/stats_c5_${lib}_pf$pf.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("   stats:", " ".join("%s %.0f" % (k, d[k]["mean"]) for k in ("solver_iterations", "stager_iterations", "order_us") if k in d))
except Exception as e:
    print("   no stats:", e)
PY
      ;;
    c5:*)
      lib=${step#c5:}
      if [ "$lib" == "-" ]; then c5 bench_c5_product --actions c5; else KAS_HIP_LIB=variants/libkas_hip_$lib.so c5 bench_c5_v_$lib --actions c5 --stats $O/stats_c5_$lib.json; fi ;;
    *) echo "unknown step $step" ;;
  esac
done
