# per-kernel totals of N = 18637 factorisations (4 repetitions) from the kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ASM_HIP_TIMING=0 rocprofv3 --kernel-trace --stats -d gpurun_out/cks -o s --output-format csv -- python3 scripts/probe/chol_time.py ${1:-18637} > gpurun_out/cks.log 2>&1
python3 - <<'PY'
import csv, glob
for fn in glob.glob("gpurun_out/cks/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(fn)))[:8]:
        print("%-60s %8s %12.1f ms %10.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/cks
