# quick FETCH_SIZE pass of bench c4 under a setting: bash pmcq.sh "<ENV=..>" tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcq_$2; rm -rf $OUT; mkdir -p $OUT
env $1 ASM_HIP_TIMING=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o f --output-format csv -- python3 bench.py --workload c4 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/log 2>&1
python3 - $OUT <<'PY'
import csv, collections, sys
d = sys.argv[1]
fs = collections.defaultdict(float); n = collections.Counter(); dur = collections.defaultdict(float)
for r in csv.DictReader(open(d + "/f_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if r["Counter_Name"] == "FETCH_SIZE": fs[k] += float(r["Counter_Value"]); n[k] += 1
for r in csv.DictReader(open(d + "/f_kernel_trace.csv")):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k in sorted(fs, key=lambda k: -fs[k])[:12]:
    print("%-28s launches %5d  fetch %8.1f MB/launch (x2 applied)  %.1f us/launch" % (k[:28], n[k], 2 * fs[k] * 1024 / n[k] / 1e6, dur[k] / n[k] / 1e3))
PY
rm -f $OUT/f_counter_collection.csv $OUT/f_kernel_trace.csv
