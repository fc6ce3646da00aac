# timeline of one N = 18637 factorisation: k_chol_panel / k_syrk_upd launches (start offset, duration) from the kernel trace.  usage: chol_trace.sh <tag> [lib]
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ASM_LIB=$2 ASM_HIP_TIMING=0 rocprofv3 --kernel-trace -d gpurun_out/ct_$TAG -o s --output-format csv -- python3 scripts/probe/chol_time.py 18637 > gpurun_out/ct_$TAG.log 2>&1
python3 - $TAG <<'PY'
import csv, glob, sys
tag = sys.argv[1]
rows = []
for fn in glob.glob("gpurun_out/ct_%s/*kernel_trace.csv" % tag):
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort()
# last factorisation: find the last k_diag_prepare
idx = max(i for i, r in enumerate(rows) if r[2].startswith("k_diag_prepare"))
t0 = rows[idx][0]
with open("gpurun_out/chol_trace_%s.txt" % tag, "w") as f:
    for s, e, n, g in rows[idx:idx + 400]:
        if n.startswith(("k_chol_panel", "k_syrk_upd")): f.write("%10.1f %10.1f %-14s %6d\n" % ((s - t0) / 1e3, (e - s) / 1e3, n[:14], g))
print(open("gpurun_out/chol_trace_%s.txt" % tag).read()[:2600])
PY
rm -rf gpurun_out/ct_$TAG
