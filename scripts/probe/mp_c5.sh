for cfg in "2 2" "4 1" "4 2" "6 1"; do
  set -- $cfg; NP=$1; CC=$2
  t0=$(date +%s.%N)
  pids=""
  for i in $(seq 1 $NP); do
    python bench.py --workload c5 --steps 12 --concurrency $CC > gpurun_out/mp_${NP}_${CC}_$i.log 2>&1 &
    pids="$pids $!"
  done
  for p in $pids; do wait $p; done
  t1=$(date +%s.%N)
  python - <<PY
import json,glob
tot=0
for f in sorted(glob.glob("gpurun_out/mp_${NP}_${CC}_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); tot+=d["value"]; print("  ",f,round(d["value"],3),d.get("batch_stats",{}).get("converged"))
print("procs",$NP,"streams",$CC,"sum solves/s",round(tot,3),"wall",round($t1-$t0,1))
PY
done
