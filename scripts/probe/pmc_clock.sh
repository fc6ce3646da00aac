# effective shader clock of each kernel (MI355X_MICROARCH.md, DVFS give-back: GRBM_GUI_ACTIVE / 8 / kernel wall time) and the
# L2-miss traffic of the same launches, on four M = 18637 factorisations.   usage: bash scripts/probe/pmc_clock.sh <tag>
TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 scripts/probe/chol_time.py 18637"
ASM_HIP_TIMING=0 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace -d gpurun_out/pmc_clk_$TAG -o c --output-format csv -- $CMD > gpurun_out/pmc_clk_$TAG.log 2>&1
echo clock rc=$?
ASM_HIP_TIMING=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_clkf_$TAG -o f --output-format csv -- $CMD > gpurun_out/pmc_clkf_$TAG.log 2>&1
echo fetch rc=$?
python3 - $TAG <<'PY'
import csv, collections, sys, glob
tag = sys.argv[1]
def load(pre):
    ctr = collections.defaultdict(dict)
    for fn in glob.glob(pre + "*counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            ctr[r["Dispatch_Id"]][r["Counter_Name"]] = ctr[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            ctr[r["Dispatch_Id"]]["_k"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    dur = {}
    for fn in glob.glob(pre + "*kernel_trace.csv"):
        for r in csv.DictReader(open(fn)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return ctr, dur
ctr, dur = load("gpurun_out/pmc_clk_%s/" % tag)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for d, c in ctr.items():
    if d not in dur: continue
    a = agg[c["_k"]]; a[0] += 1; a[1] += dur[d]; a[2] += c.get("GRBM_GUI_ACTIVE", 0.0); a[3] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[4] += c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
fctr, fdur = load("gpurun_out/pmc_clkf_%s/" % tag)
fa = collections.defaultdict(lambda: [0, 0.0])
for d, c in fctr.items():
    fa[c["_k"]][0] += 1; fa[c["_k"]][1] += 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
with open("gpurun_out/pmc_clock_%s.txt" % tag, "w") as f:
    f.write("%-28s %8s %10s %10s %12s %10s %14s\n" % ("kernel", "launches", "time_ms", "clock_GHz", "mfma_busy", "tflops", "fetch_MB/launch"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        clk = a[2] / 8.0 / a[1]                      # cycles per ns = GHz
        busy = a[3] / (a[2] / 8.0 * 1024.0) if a[2] else 0.0     # share of the cycles the chip actually ran
        f.write("%-28s %8d %10.1f %10.3f %12.3f %10.2f %14.1f\n" % (k[:28], a[0], a[1] / 1e6, clk, busy, a[4] * 512 / a[1] / 1e3, fa[k][1] / max(fa[k][0], 1) / 1e6))
print(open("gpurun_out/pmc_clock_%s.txt" % tag).read())
PY
rm -rf gpurun_out/pmc_clk_$TAG gpurun_out/pmc_clkf_$TAG
