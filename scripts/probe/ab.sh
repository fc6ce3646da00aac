# A/B of two builds of the library in ONE GPU call (boxes differ by a few per cent): bash scripts/probe/ab.sh <workloads...>
# runs every workload alternately with libasmhip_base.so (ASM_HIP_LIB) and libasmhip.so, twice each
for w in "$@"; do
  for rep in 1 2; do
    if [ $rep = 1 ]; then order="libasmhip_base.so libasmhip.so"; else order="libasmhip.so libasmhip_base.so"; fi      # (order alternates: the second run of a pair sees a warmer chip)
    for lib in $order; do
      ASM_HIP_LIB=$lib timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/ab_${w}_${lib}_$rep.log 2>&1 || exit 1
      python - "$w" "$lib" gpurun_out/ab_${w}_${lib}_$rep.log <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith('{"metric"')][-1])
print("%-5s %-20s %8.3f %s  (%.3f ms/step)" % (sys.argv[1], sys.argv[2], d["value"], d["unit"], d["ms_per_step"]), flush=True)
PY
    done
  done
done
