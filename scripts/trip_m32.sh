#!/bin/bash
# scripts/trip_m32.sh [ROUNDS]: dword mid rows (KAS_FLAG_MID32) on the GPU — parity first (the dword cases, the seeded batches, the
# headline's exact launch), then bench.py itself on ONE box in alternation: the library's default (dword mid rows where they apply)
# against the same library with KAS_PLAN_NO_MID32 (--plan-flags 1048576: the packed 16-bit rows).  Boxes differ by a few percent,
# so only figures of one trip compare.
#   gpurun --timeout 900 -- 'bash scripts/trip_m32.sh 3'
O=gpurun_out/${TRIP:-r6m32}; mkdir -p $O
R=${1:-2}
if [ -z "$SKIP_TESTS" ]; then
  timeout 500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "dword or seeded or gfx950" > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log
  timeout 400 python -m pytest tests/test_headline_launch.py -x -q -m gpu >> $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log
  tail -4 $O/tests.log
fi
for round in $(seq 1 $R); do
  for fl in 0 1048576; do
    timeout 300 python bench.py --plan-flags $fl --no-cpu --check ${BENCH_CHECK:-0} --no-extras --repeats 3 --steps 20 --warmup 5 > $O/bench_${fl}_$round.log 2>&1
    python - $O/bench_${fl}_$round.log $fl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("{") and '"metric"' in l:
        d = json.loads(l)
        fl = d["roofline"].get("in_flight_launch", {})
        print("plan flags %-8s %.1fk scenarios/s (%s)  per launch in flight: fill %.0f us order %.0f us  %s" % (
            sys.argv[2], d["value"] / 1e3, " ".join("%.0fk" % (v / 1e3) for v in d["repeats"]["values"]),
            fl.get("fill_kernel_us", 0), fl.get("order_kernel_us", 0), "dword mid rows" if "dword mid rows" in d["roofline"]["kernel"] else "16-bit mid rows"))
PY
  done
done
