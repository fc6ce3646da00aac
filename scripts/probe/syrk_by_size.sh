# per-launch rate of the trailing updates of one M = 18637 factorisation, by launch size (grid -> tile pairs), from the kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ASM_HIP_TIMING=0 rocprofv3 --kernel-trace -d gpurun_out/sbs -o s --output-format csv -- python3 scripts/probe/chol_time.py ${1:-18637} > gpurun_out/sbs.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows = []
for fn in glob.glob("gpurun_out/sbs/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        if r["Kernel_Name"].startswith(("void k_syrk<4, 8, 16, 4>", "k_syrk_upd")):
            rows.append((int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"])))
by = collections.defaultdict(list)
for g, d, s in rows: by[g].append(d)
with open("gpurun_out/syrk_by_size.txt", "w") as f:
    f.write("%8s %6s %10s %10s %12s\n" % ("blocks", "n", "min_us", "med_us", "us/block*256"))
    for g in sorted(by, reverse=True):
        v = sorted(by[g]); f.write("%8d %6d %10.1f %10.1f %12.2f\n" % (g, len(v), v[0] / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3 / g * 256))
print(open("gpurun_out/syrk_by_size.txt").read())
PY
rm -rf gpurun_out/sbs
