"""Development probe: host-bound segments of one LP in a kernel trace - runs of launches between idle gaps of more than GAP us (a host read-back or a host-side
phase sits in each gap).  Short segments between two gaps are the ping-pong patterns worth removing.  usage: segments.py p_kernel_trace.csv [which LP] [GAP us]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
GAP = float(sys.argv[3]) if len(sys.argv) > 3 else 9.0
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_ipm_init_p")]
a, b = starts[which], starts[which + 1]
seg = []
cur = [a, a, 0.0, 0.0]
for i in range(a, b):
    s, e, n = rows[i]
    gap = (s - rows[i - 1][1]) / 1e3
    if i > a and gap > GAP:
        seg.append(cur); cur = [i, i, 0.0, gap]
    cur[1] = i; cur[2] += (e - s) / 1e3
seg.append(cur)
tot_gap = sum(g[3] for g in seg)
print("LP %d: %d launches, wall %.1f us, %d segments, idle in gaps > %.0f us: %.1f us" % (which, b - a, (rows[b][0] - rows[a][0]) / 1e3, len(seg), GAP, tot_gap))
print("%8s %6s %10s   %-28s %-28s" % ("gap us", "n", "busy us", "first launch", "last launch"))
for g in seg:
    print("%8.1f %6d %10.1f   %-28s %-28s" % (g[3], g[1] - g[0] + 1, g[2], rows[g[0]][2][:28], rows[g[1]][2][:28]))
