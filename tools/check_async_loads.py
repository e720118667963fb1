#!/usr/bin/env python3
"""tools/check_async_loads.py KERNEL.s [name-substring ...]

Checks the gfx950 assembly the compiler made of the kernels that use kasw::gload_*_async (csrc/kas_wave.h): loads
issued as inline assembly, which the compiler does not know to be in flight.  Between such a load and the kernel's own
`s_waitcnt vmcnt(0)` (kasw::wait_loads, also inline assembly) NO instruction may read or write the load's destination
register — a register copy the compiler puts there (to merge two definitions of a variable, to split a live range)
would copy a value that has not arrived.  The source is written so that it has no reason to (one unconditional request
per variable, "+v" operands); this script is the proof for a given build: forward data flow over the kernel's basic
blocks, in-flight set = union over predecessors, any mention of an in-flight register outside the two kinds of assembly
statement is reported.  TEST TOOLING (tests/test_async_loads_asm.py compiles the kernels with hipcc -S and runs it).

Exit status 0 = clean; prints one line per kernel checked.
"""
from __future__ import annotations

import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^(_Z\w+):")


def regs_of(text: str) -> set[int]:
    out: set[int] = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse_function(lines: list[str]):
    """-> blocks: list of (label, [(kind, text, lineno)]), kind in {'ins', 'aload', 'await', 'asm'}"""
    blocks = [("<entry>", [])]
    in_asm = False
    asm_buf: list[tuple[str, int]] = []
    for no, raw in lines:
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
        if raw.lstrip().startswith(";;#ASMSTART"):
            in_asm, asm_buf = True, []
            continue
        if raw.lstrip().startswith(";;#ASMEND"):
            in_asm = False
            text = " ; ".join(t for t, _ in asm_buf)
            no0 = asm_buf[0][1] if asm_buf else no
            if "global_load" in text:
                blocks[-1][1].append(("aload", text, no0))
            elif "s_waitcnt vmcnt(0)" in text:
                blocks[-1][1].append(("await", text, no0))
            elif text.strip():
                blocks[-1][1].append(("asm", text, no0))
            else:
                blocks[-1][1].append(("asmempty", "", no))     # kasw::arrived / lockstep: no instruction
            continue
        if in_asm:
            if line.strip():
                asm_buf.append((line.strip(), no))
            continue
        m = LABEL.match(line)
        if m:
            blocks.append((m.group(1), []))
            continue
        s = line.strip()
        if not s or s.startswith(".") or s.startswith(";"):
            continue
        blocks[-1][1].append(("ins", s, no))
        if s.split()[0].startswith("s_cbranch") or s.split()[0] in ("s_branch", "s_endpgm", "s_setpc_b64"):
            blocks.append(("<after line %d>" % no, []))      # (a branch ends a basic block also where no label follows)
    return blocks


def check(blocks, name: str) -> list[str]:
    index = {lab: i for i, (lab, _) in enumerate(blocks)}
    succ: list[list[int]] = []
    for i, (_, ins) in enumerate(blocks):
        out: list[int] = []
        fall = True
        for kind, text, _ in ins:
            if kind != "ins":
                continue
            op = text.split()[0]
            if op.startswith("s_cbranch"):
                tgt = text.split()[-1]
                if tgt in index:
                    out.append(index[tgt])
            elif op == "s_branch":
                tgt = text.split()[-1]
                if tgt in index:
                    out.append(index[tgt])
                fall = False
            elif op in ("s_endpgm", "s_setpc_b64"):
                fall = False
        if fall and i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)
    inflight_in: list[set[int]] = [set() for _ in blocks]
    problems: dict[tuple[int, str], str] = {}
    work = [0]
    n_loads = n_waits = 0
    seen_once = [False] * len(blocks)
    while work:
        b = work.pop()
        cur = set(inflight_in[b])
        for kind, text, no in blocks[b][1]:
            if kind == "aload":
                load = text[text.index("global_load"):].split(" ; ")[0]   # (a statement may narrow EXEC around its load)
                dst = regs_of(load.split(",")[0])
                others = regs_of(",".join(load.split(",")[1:]))
                bad = (others | dst) & cur
                if bad:
                    problems[(no, text)] = f"line {no}: request touches in-flight v{sorted(bad)}: {text}"
                cur |= dst
                if not seen_once[b]:
                    n_loads += 1
            elif kind == "await":
                cur.clear()
                if not seen_once[b]:
                    n_waits += 1
            elif kind in ("asm", "ins"):
                bad = regs_of(text) & cur
                if bad:
                    problems[(no, text)] = f"line {no}: touches in-flight v{sorted(bad)}: {text}"
                if kind == "ins" and text.split()[0] in ("s_swappc_b64", "s_call_b64") and cur:
                    problems[(no, text)] = f"line {no}: call with loads in flight v{sorted(cur)}"
                if kind == "ins" and text.split()[0] == "s_endpgm" and cur:
                    problems[(no, text)] = f"line {no}: s_endpgm with loads in flight v{sorted(cur)}"
        seen_once[b] = True
        for t in succ[b]:
            if not cur <= inflight_in[t] or not seen_once[t]:
                inflight_in[t] |= cur
                work.append(t)
    msgs = sorted(problems.values())
    print(f"{name}: {n_loads} async load statements, {n_waits} waits, {len(msgs)} problems")
    return msgs


def main(argv: list[str]) -> int:
    path, wanted = argv[1], argv[2:]
    funcs: dict[str, list[tuple[int, str]]] = {}
    cur = None
    for no, line in enumerate(open(path), 1):
        m = FUNC.match(line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is not None:
            funcs[cur].append((no, line.rstrip("\n")))
    rc = 0
    checked = 0
    for name, lines in funcs.items():
        if wanted and not any(w in name for w in wanted):
            continue
        blocks = parse_function(lines)
        if not any(k == "aload" for _, ins in blocks for k, _, _ in ins):
            continue
        checked += 1
        for msg in check(blocks, name):
            print("  " + msg)
            rc = 1
    if checked == 0:
        print("no kernel with async loads found")
        return 2
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv))
