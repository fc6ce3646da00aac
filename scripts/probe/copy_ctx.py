"""Development probe: which launches precede the runtime's copy kernels in a kernel trace (rocprofv3 --kernel-trace CSV), with the idle time in front of each copy.
usage: copy_ctx.py p_kernel_trace.csv"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
ctx = collections.defaultdict(lambda: [0, 0.0, 0.0])
nmerit = sum(1 for r in rows if r[2].startswith("k_slp_merit"))
for i, (s, e, n) in enumerate(rows):
    if "copyBuffer" in n or "fillBuffer" in n:
        prev = rows[i - 1][2] if i else "-"
        nxt = rows[i + 1][2] if i + 1 < len(rows) else "-"
        c = ctx[(n[:28], prev[:34], nxt[:34])]
        c[0] += 1; c[1] += (s - rows[i - 1][1]) / 1e3 if i else 0.0
        c[2] += (rows[i + 1][0] - e) / 1e3 if i + 1 < len(rows) else 0.0
print("k_slp_merit launches (~6 per LP): %d" % nmerit)
print("%-28s %-34s %-34s %6s %10s %10s" % ("copy", "previous launch", "next launch", "count", "idle before", "idle after"))
for k, v in sorted(ctx.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print("%-28s %-34s %-34s %6d %10.1f %10.1f" % (k[0], k[1], k[2], v[0], v[1], v[2]))
