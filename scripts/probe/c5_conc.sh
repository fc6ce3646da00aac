for c in ${CONCS:-2 4 6}; do
  python bench.py --workload c5 --gpus 1 --steps 12 --warmup 1 --concurrency $c 2>/dev/null | tail -1 > gpurun_out/c5_conc_$c.json
  python - <<PY
import json
d = json.load(open("gpurun_out/c5_conc_$c.json"))
print("concurrency $c solves/s %.3f LP/s %.1f converged %d" % (d["value"], d["batch_stats"]["lp_solves"] / d["batch_stats"]["wall_s"], d["batch_stats"]["converged"]))
PY
done
