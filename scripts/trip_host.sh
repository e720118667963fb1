#!/bin/bash
# end_to_end leg of bench.py (the host-buffer boundary) for builds of the library
O=gpurun_out/$1; shift; mkdir -p $O
for lib in "$@"; do
  b=$(basename $lib .so)
  KAS_HIP_LIB=$lib timeout 600 python bench.py --no-cpu --check 0 --repeats 1 --steps 8 > $O/bench_$b.log 2>&1
  python - $O/bench_$b.log $b <<'PY'
import json, sys
line = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
e = line.get("end_to_end", {})
p = e.get("plain_every_scenario_its_own_tables", {})
print(sys.argv[2], "what-if %.0f/s" % e.get("value", 0), "plain pageable", p.get("pageable"), "pinned", p.get("pinned"))
PY
done
