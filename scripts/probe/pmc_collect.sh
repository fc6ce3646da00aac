cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
ASM_HIP_TIMING=0 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace -d gpurun_out/pmc_r02_mfma -o m --output-format csv -- $CMD > gpurun_out/pmc_r02_mfma.log 2>&1
echo mfma rc=$?
ASM_HIP_TIMING=0 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d gpurun_out/pmc_r02_clock -o c --output-format csv -- $CMD > gpurun_out/pmc_r02_clock.log 2>&1
echo clock rc=$?
ASM_HIP_TIMING=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_r02_fetch -o f --output-format csv -- $CMD > gpurun_out/pmc_r02_fetch.log 2>&1
echo fetch rc=$?
ASM_HIP_TIMING=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_r02_write -o w --output-format csv -- $CMD > gpurun_out/pmc_r02_write.log 2>&1
echo write rc=$?
python3 - <<PY
import csv, collections
for tag, pre in (("mfma","gpurun_out/pmc_r02_mfma/m"),("clock","gpurun_out/pmc_r02_clock/c"),("fetch","gpurun_out/pmc_r02_fetch/f"),("write","gpurun_out/pmc_r02_write/w")):
    try:
        rows = csv.DictReader(open(pre+"_counter_collection.csv"))
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in rows:
            k = r["Kernel_Name"].split("(")[0].replace("void ","")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            cnt[(k, r["Counter_Name"])] += 1
        out = {k: dict(v) for k, v in agg.items()}
        dur = collections.defaultdict(float); nl = collections.Counter()
        for r in csv.DictReader(open(pre+"_kernel_trace.csv")):
            k = r["Kernel_Name"].split("(")[0].replace("void ","")
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); nl[k] += 1
        for k in out:
            out[k]["_duration_ns"] = dur.get(k, 0.0); out[k]["_launches"] = nl.get(k, 0)
        import json
        json.dump({"sums": out, "dispatches": {k+"|"+c: n for (k,c),n in cnt.items()}}, open(pre+"_summary.json","w"))
        print(tag, "kernels", len(out))
    except Exception as e:
        print(tag, "ERR", e)
PY
rm -f gpurun_out/pmc_r02_*/?_counter_collection.csv gpurun_out/pmc_r02_*/?_kernel_trace.csv
