#!/bin/bash
# wave-instructions per launch of the two kernels, library by library (rocprofv3 --pmc over tools/ab_harness, one batch alone)
O=gpurun_out/${1:-insts}; shift; mkdir -p $O
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for v in "$@"; do
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --kernel-include-regex "kas_fill|kas_order" --output-format csv -d $R/$O/pmc_$v -o p -- $R/tools/ab_harness c3mix 1000 2 $R/variants/libkas_hip_$v.so > $R/$O/pmc_$v.log 2>&1
  python3 - $R/$O/pmc_$v $v <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "fill" if "kas_fill" in r["Kernel_Name"] else ("order" if "kas_order" in r["Kernel_Name"] else None)
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("fill", "order"):
    print(sys.argv[2], k, " ".join("%s %.1fM" % (c.replace("SQ_INSTS_", ""), sum(v) / len(v) / 1e6) for c, v in sorted(agg[k].items())))
PY
done
