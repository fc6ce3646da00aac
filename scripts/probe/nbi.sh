for t in 512 768 1024; do
  echo "== ASM_NBI $t"
  ASM_NBI=$t timeout -k 10 200 python scripts/probe/chol_time.py 1024 1725 3889 11192 18637 2>&1 | grep -v amdgpu
done
