#!/bin/bash
# scripts/trip_m32_parts.sh: where do the dword mid rows pay?  Tuning builds (scripts/build_variant.sh): the fill + first fit alone
# (skipord: -DKAS_TUNE_SKIP_ORDER) and the order kernel alone (ordonly: -DKAS_TUNE_ORDER_ONLY -DKAS_TUNE_NO_ROW_STORES: the fill runs
# in the first solve only and the mid rows stay), each saturated with twelve batches in flight, with and without KAS_PLAN_NO_MID32.
O=gpurun_out/${TRIP:-r6m32p}; mkdir -p $O
for round in 1 2; do
for v in skipord ordonly; do
  for fl in 0 1048576; do
    KAS_HIP_LIB=variants/libkas_hip_$v.so timeout 200 python bench.py --plan-flags $fl --no-cpu --check 0 --no-extras --repeats 3 --steps 40 --warmup 5 > $O/bench_${v}_${fl}_$round.log 2>&1
    python - $O/bench_${v}_${fl}_$round.log $v $fl <<'PY'
import json, sys
ok = False
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("{") and '"metric"' in l:
        d = json.loads(l); ok = True
        print("%-8s plan flags %-8s %.3f ms per step (%s)" % (sys.argv[2], sys.argv[3], d["ms_per_step"], " ".join("%.3f" % v for v in d["repeats"]["ms_per_step_each"])))
if not ok:
    print(sys.argv[2], sys.argv[3], "no line:", open(sys.argv[1]).read()[-600:])
PY
  done
done
done
