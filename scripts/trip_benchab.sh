#!/bin/bash
# scripts/trip_benchab.sh NAME LIB [LIB ...]: bench.py itself (eight DISTINCT batches in flight, the driver's steps) through
# builds of the library on ONE box, twice each in alternation: boxes differ by a few percent, libraries must not be
# compared across them.  BENCH_ARGS: further bench.py arguments (e.g. --plan-flags 32).
O=gpurun_out/$1; shift; mkdir -p $O
for round in 1 2; do
  for lib in "$@"; do
    b=$(basename $lib .so)
    KAS_HIP_LIB=$lib timeout 300 python bench.py $BENCH_ARGS --no-cpu --check ${BENCH_CHECK:-0} --no-extras --repeats 3 --steps 20 --warmup 5 > $O/bench_${b}_$round.log 2>&1
    python - $O/bench_${b}_$round.log $b <<'PY'
import json, sys
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("{") and '"metric"' in l:
        d = json.loads(l)
        fl = d["roofline"].get("in_flight_launch", {})
        print("%-28s %.1fk scenarios/s (%s)  per launch in flight: fill %.0f us order %.0f us" % (
            sys.argv[2], d["value"] / 1e3, " ".join("%.0fk" % (v / 1e3) for v in d["repeats"]["values"]),
            fl.get("fill_kernel_us", 0), fl.get("order_kernel_us", 0)))
PY
  done
done
