#!/usr/bin/env python
"""Per-pipe utilisation of the kernels of a solve (fill, first fit, order) (profiles/<tag>_pipe_utilisation.csv) from the SQ counter passes
of gpu_trip.sh `sq` (profiles/<tag>_pmc_sq_counters.csv), the kernel trace (one batch alone) and the bench line.
MEASUREMENT TOOLING.  usage: pipe_table.py TAG BENCH_LOG ISSUE_PROBE_LOG

Units (MI355X_MICROARCH.md, rocprofv3 PMC): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over
waves; SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count LDS-array cycles summed over CUs; SQ_INSTS_* count wave-instructions;
GRBM_GUI_ACTIVE counts cycles summed over the 8 XCDs.  Windows: one batch alone = the kernel's own duration (kernel
trace); eight batches in flight = ms_per_step of the bench line (one launch of each kernel per step, the kernels of
different steps overlap), so the in-flight rows give the utilisation of the whole job.
Capacities: 1024 SIMDs x window cycles for the issue pipes; 256 CUs x window cycles for the LDS array.  Issue cost per
wave-instruction from tools/issue_probe (same device): VALU 2.1 cycles (add/and/xor/sub/mov/lshr/bitop3) or 4.2 (the
rest: shifts left, bfe, min/max/med3, mul, mad, cmp, cndmask_e64, perm, mbcnt, readlane), SALU 4.2 per SIMD."""
import collections
import csv
import json
import sys

tag, bench_log = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f"profiles/{tag}_pmc_sq_counters.csv")))
c = collections.defaultdict(dict)
for r in rows:
    k = "fill" if "fill" in r["kernel"] else ("p4" if "kas_p4" in r["kernel"] else ("order" if "order_relax" in r["kernel"] or "order_ticket" in r["kernel"] else None))
    if k:                                                    # (a kind's kernels add up: the slim fill kernel + the full one behind it)
        d = c[(int(r["batches_in_flight"]), k)]
        d[r["counter"]] = d.get(r["counter"], 0.0) + float(r["avg_value_per_dispatch"])
dur = {}
for r in csv.DictReader(open(f"profiles/{tag}_kernel_trace_stats_one_batch_in_flight.csv")):
    if "kas_fill" in r["Name"]:
        dur["fill"] = dur.get("fill", 0.0) + float(r["AverageNs"]) * 1e-9
    if "kas_p4" in r["Name"]:
        dur["p4"] = float(r["AverageNs"]) * 1e-9
    if "kas_order_relax" in r["Name"] or "kas_order_ticket" in r["Name"]:
        dur["order"] = float(r["AverageNs"]) * 1e-9
line = json.loads([l for l in open(bench_log) if l.startswith("{")][-1])
step = line["ms_per_step"] * 1e-3
SIMDS, CUS = 1024, 256
out = [["batches_in_flight", "kernel", "window_us", "clock_GHz", "quantity", "value", "unit", "fraction_of_capacity", "how"]]


def add(mode, k, window, clock, name, value, unit, frac, how):
    out.append([mode, k, f"{window * 1e6:.1f}", f"{clock / 1e9:.2f}", name, f"{value:.4g}", unit, "" if frac is None else f"{frac:.3f}", how])


modes = sorted({m for (m, _k) in c})                        # 1 and the bench's default number of batches in flight
for mode in modes:
    each = [kk for kk in ("fill", "p4", "order") if (mode, kk) in c]     # (p4: first fit in its own kernel, round 5)
    kernels = each if mode == 1 else each + ["fill+order"]               # "fill+order": every kernel of a solve
    for k in kernels:
        if k == "fill+order":
            d = collections.Counter()
            for kk in each:
                d.update(c[(mode, kk)])
            window = step
            # (the clock of the one-batch-alone passes: in flight a kernel's duration differs from run to run, and the
            # counter pass and the bench log are two runs)
            clock = (c[(1, "fill")]["GRBM_GUI_ACTIVE"] / 8 / dur["fill"] + c[(1, "order")]["GRBM_GUI_ACTIVE"] / 8 / dur["order"]) / 2
        else:
            d = c[(mode, k)]
            window = dur[k] if mode == 1 else step
            clock = c[(1, k)]["GRBM_GUI_ACTIVE"] / 8 / dur[k]
        simd_cycles = SIMDS * window * clock
        cu_cycles = CUS * window * clock
        add(mode, k, window, clock, "waves resident per SIMD (average)", 4 * d["SQ_WAVE_CYCLES"] / simd_cycles, "waves", None, "4 x SQ_WAVE_CYCLES / SIMD-cycles")
        add(mode, k, window, clock, "wave time parked (s_waitcnt)", d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], "share", None, "SQ_WAIT_ANY / SQ_WAVE_CYCLES")
        add(mode, k, window, clock, "wave time issue-stalled", d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], "share", None, "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES")
        add(mode, k, window, clock, "VALU pipe busy, counter", 4 * d["SQ_ACTIVE_INST_VALU"], "SIMD-cycles", 4 * d["SQ_ACTIVE_INST_VALU"] / simd_cycles, "4 x SQ_ACTIVE_INST_VALU / SIMD-cycles")
        add(mode, k, window, clock, "VALU pipe busy, instructions x 2.1 cycles (lower bound)", 2.1 * d["SQ_INSTS_VALU"], "SIMD-cycles", 2.1 * d["SQ_INSTS_VALU"] / simd_cycles, "SQ_INSTS_VALU x cheapest probe cost")
        add(mode, k, window, clock, "VALU pipe busy, instructions x 4.2 cycles (upper bound)", 4.2 * d["SQ_INSTS_VALU"], "SIMD-cycles", 4.2 * d["SQ_INSTS_VALU"] / simd_cycles, "SQ_INSTS_VALU x dearest probe cost")
        add(mode, k, window, clock, "SALU issue busy", 4.2 * d["SQ_INSTS_SALU"], "SIMD-cycles", 4.2 * d["SQ_INSTS_SALU"] / simd_cycles, "SQ_INSTS_SALU x 4.2 (one scalar issue per 4.2 cycles and SIMD)")
        add(mode, k, window, clock, "LDS array busy", d["SQ_LDS_IDX_ACTIVE"], "CU-cycles", d["SQ_LDS_IDX_ACTIVE"] / cu_cycles, "SQ_LDS_IDX_ACTIVE / CU-cycles")
        add(mode, k, window, clock, "LDS array busy without bank-conflict cycles", d["SQ_LDS_IDX_ACTIVE"] - d["SQ_LDS_BANK_CONFLICT"], "CU-cycles",
            (d["SQ_LDS_IDX_ACTIVE"] - d["SQ_LDS_BANK_CONFLICT"]) / cu_cycles, "(SQ_LDS_IDX_ACTIVE - SQ_LDS_BANK_CONFLICT) / CU-cycles")
        add(mode, k, window, clock, "LDS bank-conflict share", d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], "share", None, "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
        n_all = d["SQ_INSTS_VALU"] + d["SQ_INSTS_SALU"] + d["SQ_INSTS_LDS"] + d["SQ_INSTS_VMEM_RD"] + d["SQ_INSTS_VMEM_WR"] + d["SQ_INSTS_BRANCH"] + d["SQ_INSTS_SMEM"]
        add(mode, k, window, clock, "wave-instructions per launch (VALU + SALU + LDS + VMEM + branch + SMEM)", n_all, "instructions", None,
            f"VALU {d['SQ_INSTS_VALU']:.4g} SALU {d['SQ_INSTS_SALU']:.4g} LDS {d['SQ_INSTS_LDS']:.4g} VMEM {d['SQ_INSTS_VMEM_RD'] + d['SQ_INSTS_VMEM_WR']:.4g} branch {d['SQ_INSTS_BRANCH']:.4g}")
        add(mode, k, window, clock, "issue slots per SIMD-cycle", n_all / simd_cycles, "instructions / SIMD-cycle", None,
            "a wave issues at most one instruction per ~4.3-5 cycles (probe), so this / (waves resident x (1 - parked) / 4.5) is how full the waves' own issue is")
with open(f"profiles/{tag}_pipe_utilisation.csv", "w") as f:
    csv.writer(f).writerows(out)
# which kernels and sources the table belongs to: bench.py quotes it (roofline.pipes) only for the same ones
json.dump({"csv": f"profiles/{tag}_pipe_utilisation.csv", "kernel": line["roofline"]["kernel"],
           "kernel_sources_sha16": line["roofline"]["kernel_sources_sha16"], "ms_per_step_of_the_window": line["ms_per_step"],
           "batches_in_flight": max(modes)},
          open("profiles/pipe_utilisation.json", "w"), indent=1)
for r in out:
    if r[0] != 1 or (r[0] == 1 and "busy" in r[4]):
        print(" | ".join(str(x) for x in r[:8]))
