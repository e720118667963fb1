#!/usr/bin/env python3
"""scripts/stream_timeline.py TRACE_kernel_trace.csv [OUT.csv] [--skip-first N]

Per-stream timeline of the solver's kernels from a `rocprofv3 --kernel-trace` CSV (one timed region of bench.py or
tools/ab_harness): which HIP stream sits on which hardware queue, how busy each stream is, how many fill / order
kernels execute at the same time (time-weighted), and the gaps between a stream's consecutive kernels.  Written for
VERDICT r4 W2 ("on average only 4.6 of the 8 streams have a kernel executing"): MEASUREMENT TOOLING.

The window analysed is [start of the (N+1)-th solve, end of the last kernel]; --skip-first drops warm-up solves per stream.
"""
from __future__ import annotations

import csv
import sys
from collections import defaultdict


def main(argv):
    args = [a for a in argv[1:] if not a.startswith("--")]
    skip = 0
    for i, a in enumerate(argv):
        if a == "--skip-first":
            skip = int(argv[i + 1])
            args = [x for x in args if x != argv[i + 1]]
    path = args[0]
    out = args[1] if len(args) > 1 else None
    rows = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "kas_" not in n or "selftest" in n:
            continue
        kind = ("p4" if "kas_p4" in n else "fill") if ("kas_fill" in n or "kas_spread" in n or "kas_p4" in n) else ("order" if "kas_order" in n else "other")
        rows.append({"stream": r["Stream_Id"], "queue": r["Queue_Id"], "kind": kind, "name": n.split("(")[0],
                     "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])})
    by_stream = defaultdict(list)
    for r in rows:
        by_stream[r["stream"]].append(r)
    for v in by_stream.values():
        v.sort(key=lambda r: r["s"])
    # drop the first `skip` fill kernels (and what precedes the next fill) per stream
    kept = []
    for st, v in by_stream.items():
        fills = [i for i, r in enumerate(v) if r["kind"] == "fill"]
        cut = fills[skip] if len(fills) > skip else len(v)
        by_stream[st] = v[cut:]
        kept += v[cut:]
    kept.sort(key=lambda r: r["s"])
    if not kept:
        print("no solver kernels in the trace")
        return 1
    # only streams that carry repeated solves (the timed slots)
    t0 = min(r["s"] for r in kept)
    t1 = max(r["e"] for r in kept)
    span = t1 - t0
    lines = []
    lines.append(["section", "key", "value", "unit", "note"])
    lines.append(["window", "span", "%.3f" % (span / 1e6), "ms", "first kept kernel start .. last kernel end"])
    queues = defaultdict(set)
    for st, v in sorted(by_stream.items(), key=lambda kv: int(kv[0])):
        if not v:
            continue
        busy = sum(r["e"] - r["s"] for r in v)
        gaps = [v[i + 1]["s"] - v[i]["e"] for i in range(len(v) - 1)]
        def gaps_of(a, b):
            g = sorted(v[i + 1]["s"] - v[i]["e"] for i in range(len(v) - 1) if v[i]["kind"] == a and v[i + 1]["kind"] == b)
            return g[len(g) // 2] / 1e3 if g else float("nan")
        has_p4 = any(r["kind"] == "p4" for r in v)
        q = sorted({r["queue"] for r in v})
        for qq in q:
            queues[qq].add(st)
        lines.append(["stream", st, "%.3f" % (busy / span), "busy fraction",
                      "queue %s; %d kernels; median gap %s, order->next fill %.1f us" % (
                          "/".join(q), len(v),
                          ("fill->first fit %.1f us, first fit->order %.1f us" % (gaps_of("fill", "p4"), gaps_of("p4", "order"))) if has_p4
                          else "fill->order %.1f us" % gaps_of("fill", "order"), gaps_of("order", "fill"))])
    for q, sts in sorted(queues.items(), key=lambda kv: int(kv[0])):
        lines.append(["queue", q, str(len(sts)), "streams", "streams " + " ".join(sorted(sts, key=int))])
    # time-weighted concurrency
    ev = []
    for r in kept:
        ev.append((r["s"], 1, r["kind"]))
        ev.append((r["e"], -1, r["kind"]))
    ev.sort()
    cur = {"fill": 0, "p4": 0, "order": 0, "other": 0}
    hist = defaultdict(int)
    last = ev[0][0]
    for t, d, kind in ev:
        if t > last:
            hist[(cur["fill"], cur["p4"], cur["order"])] += t - last
            last = t
        cur[kind] += d
    tot = sum(hist.values())
    share = lambda pred: "%.3f" % (sum(w for k, w in hist.items() if pred(*k)) / tot)
    with_p4 = any(k[1] for k in hist)
    lines.append(["concurrency", "fill kernels executing (average)", "%.2f" % (sum(k[0] * w for k, w in hist.items()) / tot), "kernels", ""])
    if with_p4:
        lines.append(["concurrency", "first-fit kernels executing (average)", "%.2f" % (sum(k[1] * w for k, w in hist.items()) / tot), "kernels", ""])
    lines.append(["concurrency", "order kernels executing (average)", "%.2f" % (sum(k[2] * w for k, w in hist.items()) / tot), "kernels", ""])
    lines.append(["concurrency", "fill and order kernels executing", share(lambda f, p, o: f > 0 and o > 0), "fraction of time", ""])
    lines.append(["concurrency", "fill but no order kernel executing", share(lambda f, p, o: f > 0 and o == 0), "fraction of time", ""])
    lines.append(["concurrency", "order but no fill kernel executing", share(lambda f, p, o: f == 0 and o > 0), "fraction of time", ""])
    if with_p4:
        lines.append(["concurrency", "only first-fit kernels executing", share(lambda f, p, o: f == 0 and o == 0 and p > 0), "fraction of time", ""])
    lines.append(["concurrency", "nothing executing", share(lambda f, p, o: f == 0 and o == 0 and p == 0), "fraction of time", ""])
    for (f, p4, o), w in sorted(hist.items(), key=lambda kv: -kv[1])[:12]:
        lines.append(["concurrency", ("%d fill + %d first fit + %d order" % (f, p4, o)) if with_p4 else "%d fill + %d order" % (f, o),
                      "%.3f" % (w / tot), "fraction of time", ""])
    for kind in ("fill", "p4", "order"):
        d = sorted(r["e"] - r["s"] for r in kept if r["kind"] == kind)
        if d:
            lines.append(["duration", kind, "%.3f" % (sum(d) / len(d) / 1e6), "ms average",
                          "min %.3f median %.3f max %.3f over %d launches" % (d[0] / 1e6, d[len(d) // 2] / 1e6, d[-1] / 1e6, len(d))])
    n_solves = sum(1 for r in kept if r["kind"] == "fill" and "spread" not in r["name"])
    lines.append(["rate", "solves in the window", str(n_solves), "solves", "%.3f ms per solve" % (span / 1e6 / max(n_solves, 1))])
    w = csv.writer(sys.stdout)
    for l in lines:
        w.writerow(l)
    if out:
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            for l in lines:
                w.writerow(l)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
