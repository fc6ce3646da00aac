cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/probe/vendor_dgemm.py
rocprofv3 --kernel-trace --stats -d gpurun_out/vd -o v --output-format csv -- python3 scripts/probe/vendor_dgemm.py > gpurun_out/vd.log 2>&1
python3 - <<'PY'
import csv, glob
for fn in glob.glob("gpurun_out/vd/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(fn)))[:6]:
        print(r["Name"][:230], r["Calls"], r["TotalDurationNs"])
PY
rm -rf gpurun_out/vd
