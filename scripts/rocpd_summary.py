#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) result: per-kernel calls / total / average / min / max
duration, the same table `rocprofv3 --stats` prints.  Usage: rocpd_summary.py in.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
    for name, calls, tot, avg, mn, mx in rows:
        out.write('"%s",%d,%d,%.1f,%.3f,%d,%d\n' % (name.replace('"', "'"), calls, tot, avg, 100.0 * tot / total, mn, mx))
    pmc = db.execute("select count(*) from pmc_events").fetchone()[0]
    if pmc:
        out.write("\n# PMC counters: sum per (kernel, counter) and dispatch count\n")
        q = db.execute("select * from counters_collection limit 1")
        cols = [d[0] for d in q.description]
        out.write("# columns available: %s\n" % ",".join(cols))
    return 0


if __name__ == "__main__":
    sys.exit(main())
