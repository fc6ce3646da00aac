#!/usr/bin/env python3
"""Text summary of a rocprofv3 --kernel-trace --stats SQLite database (the format committed under profiles/).
usage: prof_summary.py <results.db> <bench-log> "<header line>" > profiles/<name>.txt"""
import sqlite3, sys, re
db, log, header = sys.argv[1], sys.argv[2], sys.argv[3]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), min(start), max(end) from kernels group by name order by 3 desc").fetchall()
span0 = min(r[3] for r in rows); span1 = max(r[4] for r in rows)
tot = sum(r[2] for r in rows); n = sum(r[1] for r in rows)
print("# " + header)
print("# MI355X, round 1")
line = [l for l in open(log) if l.startswith('{"metric"')]
if line:
    print("# bench line printed by the profiled run:")
    print("# " + line[-1].strip())
print("# kernel span %.1f ms, sum of kernel durations %.1f ms, %d launches" % ((span1 - span0) / 1e6, tot / 1e6, n))
print("%-115s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, cnt, t, _, _ in rows:
    print("%-115s %8d %14.1f %12.3f %8.2f" % (name[:115], cnt, t / 1e3, t / cnt / 1e3, 100.0 * t / tot))
