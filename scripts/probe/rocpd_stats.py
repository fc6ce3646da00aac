"""Per-kernel summary of a rocprofv3 --kernel-trace run stored as a rocpd database (the default output format of this rocprofv3):
   python scripts/probe/rocpd_stats.py <results.db> [top]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cur = db.cursor()
rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), avg(1.0*grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z)) "
                        "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
span = list(cur.execute("select min(start), max(end) from kernels"))[0]
print("kernel time %.1f ms over a span of %.1f ms, %d launches" % (tot / 1e6, (span[1] - span[0]) / 1e6, sum(r[1] for r in rows)))
for r in rows[:top]:
    nm = re.sub(r"\(.*", "", r[0])[:64]
    print("%-64s calls %7d  total %8.1f ms  avg %7.1f us  workgroups %7.0f  %5.1f%%" % (nm, r[1], r[2] / 1e6, r[3] / 1e3, r[4], 100 * r[2] / tot))
