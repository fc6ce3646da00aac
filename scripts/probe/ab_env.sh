# A/B of environment settings in ONE GPU call: bash scripts/probe/ab_env.sh <workload> "<ENV=.. ENV=..>" ["<other setting>" ...]   (first = baseline "")
w=$1; shift
for rep in 1 2; do
  for setting in "" "$@"; do
    env $setting timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/abenv.log 2>&1 || exit 1
    python - "$w" "$setting" gpurun_out/abenv.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith('{"metric"')][-1])
print("%-5s %-40s %8.3f %s  (%.3f ms/step)" % (sys.argv[1], sys.argv[2] or "(default)", d["value"], d["unit"], d["ms_per_step"]), flush=True)
PY
  done
done
