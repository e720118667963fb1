#!/bin/bash
# scripts/trip_sizes.sh: bench.py (distinct batches in flight) at the two tile sizes of the relaxation form, in alternation on one box
O=gpurun_out/sz; mkdir -p $O
run() { n=$1; shift; timeout 300 python bench.py --no-cpu --check 0 --no-extras --repeats 3 --steps 20 --warmup 5 "$@" > $O/$n.log 2>&1; python - $O/$n.log $n <<'PY'
import json, sys
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("{") and '"metric"' in l:
        d = json.loads(l)
        print("%-22s %.1fk scenarios/s (%s) ms/step %.3f" % (sys.argv[2], d["value"] / 1e3, " ".join("%.0fk" % (v / 1e3) for v in d["repeats"]["values"]), d["ms_per_step"]))
PY
}
for i in 1 2 3; do
run t64_$i --plan-flags 131072
run t128_$i --plan-flags 262144
done
run t128x12 --plan-flags 262144 --in-flight 12
run t128x6 --plan-flags 262144 --in-flight 6
run t128_long --plan-flags 262144 --steps 40
