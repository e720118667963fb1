#!/bin/bash
# round 2, trip H: SQ counters of the order kernel, before (orig) and after (cur) the mid rows
set -x
O=gpurun_out/r2h
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in orig cur; do
  KAS_HIP_LIB=$GRAFT_REPO_ROOT/variants/libkas_hip_$v.so timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/$O/sq1_$v -o sq -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 3 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/sq1_$v.log 2>&1; echo "sq1 $v exit $?"
  KAS_HIP_LIB=$GRAFT_REPO_ROOT/variants/libkas_hip_$v.so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/sq2_$v -o sq -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 3 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/sq2_$v.log 2>&1; echo "sq2 $v exit $?"
  KAS_HIP_LIB=$GRAFT_REPO_ROOT/variants/libkas_hip_$v.so timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $GRAFT_REPO_ROOT/$O/sq3_$v -o sq -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --check 0 --no-extras --steps 3 --warmup 1 --in-flight 1 > $GRAFT_REPO_ROOT/$O/sq3_$v.log 2>&1; echo "sq3 $v exit $?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for v in ("orig", "cur"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r2h/sq*_{v}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "kas_" in r["Kernel_Name"]:
                k = "fill" if "fill" in r["Kernel_Name"] else ("perm" if "perm" in r["Kernel_Name"] else "order")
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), vals in sorted(agg.items()):
        if k != "perm": print(v, k, c, f"{sum(vals)/len(vals):.4g}", len(vals))
PY
