#!/usr/bin/env python3
"""Text summary of a rocprofv3 --kernel-trace --stats run (CSV output) in the format committed under profiles/.
usage: prof_summary.py <dir>/<prefix> <bench-log> "<header line>" > profiles/<name>.txt
reads <prefix>_kernel_stats.csv and, for the kernel span, <prefix>_kernel_trace.csv[.gz]"""
import csv, gzip, os, sys
prefix, log, header = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(prefix + "_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = sum(int(r["Calls"]) for r in rows)
span = None
tr = prefix + "_kernel_trace.csv"
op = gzip.open(tr + ".gz", "rt") if os.path.exists(tr + ".gz") else (open(tr) if os.path.exists(tr) else None)
if op is not None:
    s, e = None, None
    for r in csv.DictReader(op):
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        s = a if s is None or a < s else s
        e = b if e is None or b > e else e
    span = (e - s) / 1e6
print("# " + header)
line = [l for l in open(log) if l.startswith('{"metric"')]
if line:
    print("# bench line printed by the profiled run:")
    print("# " + line[-1].strip())
print("# kernel span %s ms, sum of kernel durations %.1f ms, %d launches" % ("%.1f" % span if span else "n/a", tot / 1e6, n))
print("%-100s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for r in rows:
    t = float(r["TotalDurationNs"])
    print("%-100s %8d %14.1f %12.3f %8.2f" % (r["Name"][:100], int(r["Calls"]), t / 1e3, float(r["AverageNs"]) / 1e3, 100.0 * t / tot))
