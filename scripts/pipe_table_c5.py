#!/usr/bin/env python
"""Per-pipe table of BASELINE configs[4]'s kernels (one scenario, 1M partitions x 5k brokers x RF 5) from the SQ counter passes of
gpu_trip.sh `sqc5` (where kas_order_wide_kernel<5>'s 2.2 us per 64-row tile go).  MEASUREMENT TOOLING.
usage: pipe_table_c5.py GPURUN_DIR TAG   ->  profiles/TAG_pmc_sq_counters_config5.csv, profiles/TAG_pipe_utilisation_config5.csv

Units as scripts/pipe_table.py: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves, SQ_LDS_* LDS-array
cycles summed over CUs, SQ_INSTS_* wave-instructions, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs."""
import collections
import csv
import glob
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "prof_c5sq*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "kas_" in r["Kernel_Name"] and "selftest" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(f"profiles/{tag}_pmc_sq_counters_config5.csv", "w") as out:
    out.write("kernel,dispatches,counter,avg_value_per_dispatch\n")
    for (k, c), v in sorted(agg.items()):
        out.write(f'"{k}",{len(v)},{c},{sum(v) / len(v):.1f}\n')
dur = {}
for f in glob.glob(os.path.join(d, "prof_trace_config5", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "kas_" in r["Name"]:
            dur[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"]) * 1e-9
rows = [["kernel", "duration_us", "clock_GHz", "quantity", "value", "unit", "how"]]
for k in sorted({kk for kk, _ in agg}):
    c = {cc: sum(v) / len(v) for (kk, cc), v in agg.items() if kk == k}
    if k not in dur or "GRBM_GUI_ACTIVE" not in c or c.get("SQ_WAVE_CYCLES", 0) <= 0:
        continue
    t = dur[k]
    clock = c["GRBM_GUI_ACTIVE"] / 8 / t
    cyc = t * clock                                             # cycles of the kernel's duration

    def add(name, value, unit, how):
        rows.append([k, f"{t * 1e6:.1f}", f"{clock / 1e9:.2f}", name, f"{value:.4g}", unit, how])
    wave_cycles = 4 * c["SQ_WAVE_CYCLES"]
    add("wavefronts resident (average over the kernel's duration)", wave_cycles / cyc, "waves", "4 x SQ_WAVE_CYCLES / duration cycles")
    add("wave time parked (s_waitcnt)", c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "share", "SQ_WAIT_ANY / SQ_WAVE_CYCLES")
    add("wave time issue-stalled", c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], "share", "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES")
    add("wave time executing VALU", c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], "share", "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES")
    add("wave time executing SALU", c["SQ_ACTIVE_INST_SCA"] / c["SQ_WAVE_CYCLES"], "share", "SQ_ACTIVE_INST_SCA / SQ_WAVE_CYCLES")
    add("wave time executing LDS instructions", c["SQ_ACTIVE_INST_LDS"] / c["SQ_WAVE_CYCLES"], "share", "SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES")
    add("VALU wave-instructions", c["SQ_INSTS_VALU"], "instructions", "SQ_INSTS_VALU")
    add("SALU wave-instructions", c["SQ_INSTS_SALU"], "instructions", "SQ_INSTS_SALU")
    add("LDS wave-instructions", c["SQ_INSTS_LDS"], "instructions", "SQ_INSTS_LDS")
    add("VMEM wave-instructions (read + write)", c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"], "instructions", "SQ_INSTS_VMEM_RD + _WR")
    add("VALU issue cycles per wavefront if spread evenly", 4 * c["SQ_ACTIVE_INST_VALU"] / max(wave_cycles / cyc, 1e-9) / cyc, "share of the duration",
        "4 x SQ_ACTIVE_INST_VALU / resident waves / duration cycles")
    add("LDS array busy", c["SQ_LDS_IDX_ACTIVE"] / cyc, "CU-equivalents", "SQ_LDS_IDX_ACTIVE / duration cycles (the kernel's one workgroup sits on one CU)")
    add("LDS bank-conflict share", c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), "share", "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
    if "order_wide" in k:
        tiles = 1000000 / 64
        add("per 64-row tile: duration", cyc / tiles, "cycles", "duration cycles / 15,625 tiles")
        add("per 64-row tile: VALU wave-instructions (all five wavefronts)", c["SQ_INSTS_VALU"] / tiles, "instructions", "")
        add("per 64-row tile: LDS wave-instructions", c["SQ_INSTS_LDS"] / tiles, "instructions", "")
        add("per 64-row tile: SALU wave-instructions", c["SQ_INSTS_SALU"] / tiles, "instructions", "")
with open(f"profiles/{tag}_pipe_utilisation_config5.csv", "w") as f:
    csv.writer(f).writerows(rows)
for r in rows:
    if "order_wide" in r[0] or r[0] == "kernel":
        print(" | ".join(str(x) for x in r[:6]))
