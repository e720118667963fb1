#!/usr/bin/env python3
"""tools/isa_loops.py FILE.s KERNEL_SUBSTRING [MIN_DEPTH]

The loops of one kernel in a `hipcc -S --cuda-device-only` listing: every basic block inside a loop (LLVM's own "Depth=" remarks)
with its instruction classes AND its memory signature (which global / LDS instructions it holds, counted) — which is what
identifies a hot loop in a kernel of many instantiated paths: the fill's first row scan is the block with four
global_load_dwordx3 and twelve ds_add_u32, its second scan the one with ds_add_rtn_u32 and the mid-row stores, the relaxation
step the one with three ds_sub_u32 and three ds_add_rtn_u32.  MEASUREMENT TOOLING, not a product path."""
import collections
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_stats import classify  # noqa: E402


def main():
    path, want = sys.argv[1], sys.argv[2]
    min_depth = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z][\w$.]*:", l) and want in l and not l.startswith(".L"))
    blocks, cur = [], None
    for i in range(start + 1, len(lines)):
        l = lines[i].strip()
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB[\w$.]*):(.*)", l)
        if m:
            d = re.search(r"Depth=(\d+)", m.group(2))
            cur = {"label": m.group(1), "line": i + 1, "depth": int(d.group(1)) if d else 0, "cls": collections.Counter(),
                   "mem": collections.Counter(), "header": "Loop Header" in m.group(2)}
            blocks.append(cur)
            continue
        if cur is None or not l or l.startswith((";", ".", "//")):
            # (LLVM puts the loop remarks of a block on comment lines behind its label)
            if cur is not None and l.startswith(";"):
                d = re.search(r"Depth=(\d+)", l)
                if d and not cur["cls"]:
                    cur["depth"] = max(cur["depth"], int(d.group(1)))
                    cur["header"] = cur["header"] or "Loop Header" in l
            continue
        op = l.split()[0]
        c = classify(op)
        cur["cls"][c] += 1
        if c in ("lds", "vmem"):
            cur["mem"][op] += 1
    print(f"{lines[start].split(':')[0]}: {len(blocks)} basic blocks, {sum(sum(b['cls'].values()) for b in blocks)} instructions; "
          f"blocks inside loops of depth >= {min_depth}:")
    for b in blocks:
        n = sum(b["cls"].values())
        if b["depth"] >= min_depth and n >= 12:
            cls = " ".join(f"{k}={b['cls'][k]}" for k in ("valu", "salu", "lds", "vmem", "wait", "branch", "misc") if b["cls"].get(k))
            mem = ", ".join(f"{v} x {k}" for k, v in sorted(b["mem"].items()))
            print(f"  {b['label']:<12} line {b['line']:>6} depth {b['depth']}{' (header)' if b['header'] else ''}  n={n:<4} {cls}\n{'':16}{mem}")


if __name__ == "__main__":
    main()
