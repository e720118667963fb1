#!/bin/bash
# scripts/trip_phase.sh NAME LIB...: per-phase times of the fill kernel (kas_plan_stats, one batch alone) for builds of the library
O=gpurun_out/$1; shift; mkdir -p $O
for lib in "$@"; do
  b=$(basename $lib .so)
  KAS_HIP_LIB=$lib timeout 300 python bench.py --no-cpu --check 0 --no-extras --repeats 1 --steps 10 --in-flight 1 --stats $O/stats_$b.json > $O/bench_$b.log 2>&1
  python - $O/stats_$b.json $b <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], " ".join("%s %.0f" % (k, d[k]["mean"]) for k in ("setup_us", "p2_hist_quota_us", "p2_keep_p3_us", "p4_us", "order_us", "fill_kernel_avg_us", "order_kernel_avg_us") if k in d) if isinstance(d.get("setup_us"), dict) else d)
PY
done
