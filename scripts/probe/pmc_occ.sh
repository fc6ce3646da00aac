# mean resident wavefronts per SIMD of each kernel: SQ_WAVE_CYCLES (quad-cycles, summed over waves) x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ASM_HIP_TIMING=0 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d gpurun_out/pmc_occ_$TAG -o c --output-format csv -- python3 scripts/probe/chol_time.py 18637 > gpurun_out/pmc_occ_$TAG.log 2>&1
python3 - $TAG <<'PY'
import csv, collections, sys, glob
tag = sys.argv[1]
ctr = collections.defaultdict(dict); dur = {}
for fn in glob.glob("gpurun_out/pmc_occ_%s/*counter_collection.csv" % tag):
    for r in csv.DictReader(open(fn)):
        d = ctr[r["Dispatch_Id"]]; d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); d["_k"] = r["Kernel_Name"].split("(")[0].replace("void ", ""); d["_g"] = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
for fn in glob.glob("gpurun_out/pmc_occ_%s/*kernel_trace.csv" % tag):
    for r in csv.DictReader(open(fn)): dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = [(c["_g"], dur.get(d, 0), c) for d, c in ctr.items() if c["_k"].startswith("k_syrk_upd")]
rows.sort(key=lambda t: -t[0])
with open("gpurun_out/pmc_occ_%s.txt" % tag, "w") as f:
    f.write("%8s %10s %8s %12s %10s %10s %10s %10s\n" % ("blocks", "us", "GHz", "waves/SIMD", "mfma_busy", "wait_inst", "wait_any", "sq_busy"))
    seen = set()
    for g, du, c in rows:
        if g in seen: continue
        seen.add(g)
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        f.write("%8d %10.1f %8.3f %12.3f %10.3f %10.3f %10.3f %10.3f\n" % (g, du / 1e3, cyc / max(du, 1), wc * 4.0 / (cyc * 1024.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0),
                c.get("SQ_WAIT_INST_ANY", 0.0) / wc, c.get("SQ_WAIT_ANY", 0.0) / wc, c.get("SQ_BUSY_CYCLES", 0.0) / (cyc * 32)))
print(open("gpurun_out/pmc_occ_%s.txt" % tag).read()[:3000])
PY
rm -rf gpurun_out/pmc_occ_$TAG
