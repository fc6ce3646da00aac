# Profile set of one workload for profiles/: kernel trace + stats of the bench command, then the PMC passes, each in its own run
# (rocprofv3 serialises kernels under --pmc; counters are never combined with a trace domain other than --kernel-trace).
# usage (on the GPU box, from the repo root):  bash scripts/probe/pmc_collect.sh <round tag, e.g. r03> <workload> [steps]
TAG=${1:-r03}; WL=${2:-c4}; STEPS=${3:-5}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $ROOT
OUT=gpurun_out/pmc_${TAG}_${WL}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o p --output-format csv -- python3 bench.py --workload $WL --no-cpu-baseline > $OUT/stats.log 2>&1
echo stats rc=$?
python3 scripts/prof_summary.py $OUT/stats/p $OUT/stats.log "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $WL --no-cpu-baseline   (MI355X, round ${TAG#r}; ASM_HIP_TIMING=1)" > gpurun_out/${TAG}_${WL}_kernel_stats.txt
rm -f $OUT/stats/p_kernel_trace.csv
CMD="python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline"
ASM_HIP_TIMING=0 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace -d $OUT/mfma -o m --output-format csv -- $CMD > $OUT/mfma.log 2>&1
echo mfma rc=$?
ASM_HIP_TIMING=0 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $OUT/clock -o c --output-format csv -- $CMD > $OUT/clock.log 2>&1
echo clock rc=$?
ASM_HIP_TIMING=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f --output-format csv -- $CMD > $OUT/fetch.log 2>&1
echo fetch rc=$?
ASM_HIP_TIMING=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w --output-format csv -- $CMD > $OUT/write.log 2>&1
echo write rc=$?
python3 - $OUT <<'PY'
import csv, collections, json, sys
out_dir = sys.argv[1]
for tag, pre in (("mfma", out_dir + "/mfma/m"), ("clock", out_dir + "/clock/c"), ("fetch", out_dir + "/fetch/f"), ("write", out_dir + "/write/w")):
    try:
        rows = csv.DictReader(open(pre + "_counter_collection.csv"))
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        out = {k: dict(v) for k, v in agg.items()}
        dur = collections.defaultdict(float); nl = collections.Counter()
        for r in csv.DictReader(open(pre + "_kernel_trace.csv")):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); nl[k] += 1
        for k in out:
            out[k]["_duration_ns"] = dur.get(k, 0.0); out[k]["_launches"] = nl.get(k, 0)
        json.dump({"sums": out, "dispatches": {k + "|" + c: n for (k, c), n in cnt.items()}}, open(pre + "_summary.json", "w"))
        print(tag, "kernels", len(out))
    except Exception as e:
        print(tag, "ERR", e)
PY
rm -f $OUT/*/?_counter_collection.csv $OUT/*/?_kernel_trace.csv
python3 scripts/pmc_summaries.py $TAG $WL "$CMD"
