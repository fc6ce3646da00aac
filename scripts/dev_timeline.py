"""Timeline analysis of one Cholesky factorisation out of a rocprofv3 kernel-trace database: which kernels are on the
critical path (running alone) and how much of the span is idle."""
import sqlite3, sys, re
db = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = sqlite3.connect(db)
rows = c.execute("select name,start,end,stream_id,grid_x,workgroup_x from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")
# a factorisation = from a k_diag_prepare (or first k_potrf_diag after a k_syrk with big grid) to the k_trtri512
starts = [i for i, r in enumerate(rows) if short(r[0]).startswith("k_diag_prepare")]
ends = [i for i, r in enumerate(rows) if short(r[0]).startswith("k_trtri_init") or short(r[0]).startswith("k_trtri512")]
i0 = starts[which]; i1 = min(e for e in ends if e > i0)
seg = rows[i0:i1 + 1]
t0 = seg[0][1]; t1 = max(r[2] for r in seg)
print("factorisation %d: %d launches, span %.2f ms" % (which, len(seg), (t1 - t0) / 1e6))
ev = []
for n, s, e, st, gx, wx in seg:
    ev.append((s, 1, short(n), gx // max(wx, 1))); ev.append((e, -1, short(n), gx // max(wx, 1)))
ev.sort()
active = {}
last = t0; alone = {}; idle = 0; both = 0
for t, d, n, g in ev:
    dt = t - last
    if dt > 0:
        names = [k for k, v in active.items() if v > 0]
        if not names: idle += dt
        elif len(names) == 1: alone[names[0]] = alone.get(names[0], 0) + dt
        else: both += dt
    key = n if not n.startswith("k_syrk") else n + ("[big]" if g > 600 else "[small]")
    active[key] = active.get(key, 0) + d
    last = t
print("idle %.2f ms, >=2 kernel kinds overlapped %.2f ms" % (idle / 1e6, both / 1e6))
for k, v in sorted(alone.items(), key=lambda kv: -kv[1]): print("  alone %-28s %8.2f ms" % (k, v / 1e6))
tot = {}
for n, s, e, st, gx, wx in seg:
    key = short(n) if not short(n).startswith("k_syrk") else short(n) + ("[big]" if gx // max(wx, 1) > 600 else "[small]")
    a = tot.setdefault(key, [0, 0]); a[0] += e - s; a[1] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0]): print("  total %-28s %8.2f ms  %5d launches  avg %7.1f us" % (k, v[0] / 1e6, v[1], v[0] / v[1] / 1e3))
