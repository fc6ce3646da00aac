"""Development probe: the launches around one SLP step boundary (from the last merit evaluation of a step to the first interior-point launch of the next LP),
with start times relative to the first and the idle time in front of each.  usage: step_seq.py p_kernel_trace.csv [which]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
fn = [i for i, r in enumerate(rows) if r[2].startswith("k_fn_rows")]
i0 = fn[which]
a = i0
while a > 0 and not rows[a][2].startswith("k_chol_panel"): a -= 1      # back to the last factorisation of the previous LP
b = i0
while b < len(rows) and not rows[b][2].startswith("k_ipm_init_p"): b += 1
t0 = rows[a][0]
for i in range(a, min(b + 3, len(rows))):
    s, e, n = rows[i]
    print("%10.1f  %8.1f  idle %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - rows[i - 1][1]) / 1e3, n[:60]))
