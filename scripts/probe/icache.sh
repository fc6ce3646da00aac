# instruction-cache counters of the panel kernels (development probe): bash scripts/probe/icache.sh N[:band] ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ic; mkdir -p gpurun_out/ic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/ic -o p --output-format csv -- python3 scripts/probe/chol_time.py "$@" > gpurun_out/ic/log 2>&1
echo rc=$?
tail -3 gpurun_out/ic/log
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
try:
    for r in csv.DictReader(open("gpurun_out/ic/p_counter_collection.csv")):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQC_ICACHE_REQ", 0))[:12]:
        d = n[(k, "SQC_ICACHE_REQ")] or 1
        print("%-28s launches %4d  icache req %12.0f  hits %12.0f  misses %10.0f  (per launch: req %9.0f miss %8.0f)" % (k[:28], d, v.get("SQC_ICACHE_REQ", 0), v.get("SQC_ICACHE_HITS", 0), v.get("SQC_ICACHE_MISSES", 0), v.get("SQC_ICACHE_REQ", 0) / d, v.get("SQC_ICACHE_MISSES", 0) / d))
except Exception as e:
    print("no counters:", e)
PY
