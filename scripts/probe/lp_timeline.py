"""Where one steady-state LP spends its time (development probe): reads the kernel trace CSV of a rocprofv3 --kernel-trace run of `bench.py --workload c4`,
cuts the timeline at the LP boundaries (k_ns_zero_band = first kernel of the null-space set-up of an LP) and, inside an LP, at the interior-point
iterations (k_ipm_theta_ns; k_ns_theta before the fusion), and prints for the LAST LPs: per phase the wall span, the sum of kernel durations and the idle time (no kernel running on any
stream), plus the per-kernel busy / gap table of one LP.   usage: lp_timeline.py <prefix>_kernel_trace.csv [n_lps]"""
import csv, collections, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 6
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_ns_zero_band")]
def span_stats(seg):
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    busy, cur_e = 0, t0
    for s, e, _ in seg:                      # union of intervals (two streams overlap during the S0 factorisation)
        if e <= cur_e: continue
        busy += e - max(s, cur_e); cur_e = e
    return (t1 - t0) / 1e3, busy / 1e3
for li in range(len(starts) - nl - 1, len(starts) - 1):
    seg = rows[starts[li]:starts[li + 1]]
    wall, busy = span_stats(seg)
    # phases: set-up = up to the first k_ns_theta, ipm = first theta .. last k_ns_update/k_ipm_steps region, rest = polish + SLP
    th = [i for i, r in enumerate(seg) if (r[2].startswith("k_ns_theta") or r[2].startswith("k_ipm_theta_ns"))]
    if not th:
        print("LP %d: %.1f us wall (no null-space iterations)" % (li, wall)); continue
    last_upd = max(i for i, r in enumerate(seg) if r[2].startswith("k_ns_update"))
    a = span_stats(seg[:th[0]]); b = span_stats(seg[th[0]:last_upd + 1]); c = span_stats(seg[last_upd + 1:])
    print("LP %3d: wall %7.0f us busy %7.0f (%.0f%%) launches %d | set-up %6.0f/%6.0f | ipm %6.0f/%6.0f (%d its, %.0f us per it, %d launches per it) | polish+rest %6.0f/%6.0f (%d launches)"
          % (li, wall, busy, 100 * busy / wall, len(seg), a[0], a[1], b[0], b[1], len(th), b[0] / len(th), (last_upd + 1 - th[0]) // len(th), c[0], c[1], len(seg) - last_upd - 1))
seg = rows[starts[-2]:starts[-1]]
agg = collections.OrderedDict()
pe = seg[0][0]
for s, e, n in seg:
    d = agg.setdefault(n, [0, 0.0, 0.0])
    d[0] += 1; d[1] += (e - s) / 1e3; d[2] += max(0, s - pe) / 1e3
    pe = max(pe, e)
print("\nlast LP by kernel: calls, busy us, idle us in front of its launches")
for n, d in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:45]:
    print("  %-60s %5d %9.1f %9.1f" % (n[:60], d[0], d[1], d[2]))
# one interior-point iteration of the last LP, launch by launch
th = [i for i, r in enumerate(seg) if (r[2].startswith("k_ns_theta") or r[2].startswith("k_ipm_theta_ns"))]
if len(th) > 6:
    it = seg[th[5]:th[6]]
    print("\none iteration (launch, duration us, gap to the previous end us):")
    pe = it[0][0]
    for s, e, n in it:
        print("  %-50s %7.1f %7.1f" % (n[:50], (e - s) / 1e3, (s - pe) / 1e3)); pe = max(pe, e)
# the polish / rest segment of the last LP
last_upd = max(i for i, r in enumerate(seg) if r[2].startswith("k_ns_update"))
print("\nafter the last iteration (launch, duration us, gap us):")
pe = seg[last_upd][1]
for s, e, n in seg[last_upd + 1:]:
    print("  %-50s %7.1f %7.1f" % (n[:50], (e - s) / 1e3, (s - pe) / 1e3)); pe = max(pe, e)
