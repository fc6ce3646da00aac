# C5 throughput against the number of HIP hardware queues and of scenario streams per GPU
for cfg in "4 4" "6 4" "8 4" "8 4" "8 5" "8 6" "12 4" "12 6"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 python bench.py --workload c5 --steps 48 --concurrency $2 --no-cpu-baseline > gpurun_out/hwq_$1_$2.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/hwq_$1_$2.log"):
    if l.startswith("{"):
        d = json.loads(l); print("HWQ $1 streams $2:", round(d["value"], 3), d["unit"], d.get("batch_stats", {}).get("converged"))
PY
done
