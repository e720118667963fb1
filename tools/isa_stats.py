#!/usr/bin/env python3
"""tools/isa_stats.py FILE.s KERNEL_SUBSTRING [--blocks]

Instruction-class census of one kernel in a `hipcc -S --cuda-device-only` listing: VALU / SALU / LDS / VMEM /
branch / waitcnt per basic block (label to label), so that the instruction streams of the hot loops can be
counted before a GPU minute is spent (the kernels of this repository are VALU-issue bound: DESIGN.md section 4.5).
MEASUREMENT TOOLING, not a product path."""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_call")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith(("s_nop", "s_sleep", "s_setprio", "s_barrier")):
        return "misc"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, want = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[_A-Za-z][\w$.]*:", l) and want in l and not l.startswith(".L"):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = []
    cur = {"label": lines[start].split(":")[0], "line": start + 1}
    counts = {}
    total = {}
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if l.startswith(".Lfunc_end") or l.startswith(".section"):
            break
        m = re.match(r"^(\.LBB[\w$.]*):", l)
        if m:
            cur["counts"] = counts
            blocks.append(cur)
            cur = {"label": m.group(1), "line": i + 1}
            counts = {}
            continue
        if not l or l.startswith((";", ".", "//")):
            continue
        op = l.split()[0]
        c = classify(op)
        counts[c] = counts.get(c, 0) + 1
        total[c] = total.get(c, 0) + 1
        if c == "valu":
            # 64-bit / multi-pass VALU ops are worth noting
            for k in ("_b64", "_u64", "_i64", "mul_lo", "mul_hi", "mad_u64", "mad_i64"):
                if k in op:
                    counts["valu64"] = counts.get("valu64", 0) + 1
                    total["valu64"] = total.get("valu64", 0) + 1
                    break
    cur["counts"] = counts
    blocks.append(cur)
    keys = ["valu", "valu64", "salu", "lds", "vmem", "smem", "wait", "branch", "misc", "other"]
    print("total", " ".join("%s=%d" % (k, total.get(k, 0)) for k in keys))
    if show_blocks:
        for b in blocks:
            n = sum(v for k, v in b["counts"].items() if k != "valu64")
            if n >= 8:
                print("%-12s line %6d  n=%4d  %s" % (b["label"], b["line"], n,
                      " ".join("%s=%d" % (k, b["counts"][k]) for k in keys if b["counts"].get(k))))


if __name__ == "__main__":
    main()
