for c in 6 12 19; do
  for w in c4 c3; do
    echo "== COL_MAX_CG $c $w"
    ASM_COL_MAX_CG=$c timeout -k 10 300 python bench.py --workload $w --steps 6 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), d['config']['factorisations_per_step'], d['lp_outcomes']['paths'])"
  done
done
