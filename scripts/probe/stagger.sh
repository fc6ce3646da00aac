for s in 0 4 2; do
  echo "== stagger $s"
  ASM_UPD_STAGGER=$s timeout -k 10 200 python scripts/probe/chol_time.py 5000 11192 18637 2>&1 | grep -v amdgpu
done
ASM_UPD_STAGGER=4 timeout -k 10 300 bash scripts/probe/pmc_clock.sh stg4 > /dev/null; head -3 gpurun_out/pmc_clock_stg4.txt
