"""Scratch: kernel sequence of a rocprofv3 --kernel-trace database (rocpd sqlite): name, grid (workgroups), duration, gap to the previous kernel."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
names = dict(db.execute("select id, kernel_name from %s" % ks).fetchall())
rows = db.execute("select kernel_id, start, end, grid_size_x, workgroup_size_x, grid_size_y, workgroup_size_y, stream_id from %s order by start" % kd).fetchall()
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(rows)
pe = None
for i, r in enumerate(rows):
    nm = re.sub(r"^_Z\d+", "", names[r[0]].split(".kd")[0])[:34]
    if lo <= i < hi:
        print("%6d %-34s g%-5d y%-4d s%-3s %8.1f us  gap %7.1f" % (i, nm, r[3] // max(r[4], 1), r[5] // max(r[6], 1), r[7], (r[2] - r[1]) / 1e3, (r[1] - pe) / 1e3 if pe else 0))
    pe = max(pe or 0, r[2])
print("dispatches", len(rows))
