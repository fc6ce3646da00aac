"""One iteration launch by launch (development probe): kernel trace CSV of a rocprofv3 --kernel-trace run, cut at a marker kernel.
usage: iter_seq.py <prefix>_kernel_trace.csv <marker kernel prefix> [which occurrence]"""
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(sys.argv[1])))
mk = [i for i, r in enumerate(rows) if r[2].startswith(sys.argv[2])]
w = int(sys.argv[3]) if len(sys.argv) > 3 else len(mk) // 2
seg = rows[mk[w]:mk[w + 1]]
pe = seg[0][0]
print("%d launches, %.1f us wall, %.1f us kernels" % (len(seg), (rows[mk[w + 1]][0] - seg[0][0]) / 1e3, sum(e - s for s, e, _ in seg) / 1e3))
for s, e, n in seg:
    print("  %-46s %7.1f %7.1f" % (n[:46], (e - s) / 1e3, (s - pe) / 1e3)); pe = max(pe, e)
