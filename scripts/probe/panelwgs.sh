for p in 480 360 240 160; do
  echo "== ASM_PANEL_WGS $p"
  ASM_PANEL_WGS=$p timeout -k 10 200 python scripts/probe/chol_time.py 5000 11192 18637 2>&1 | grep -v amdgpu
done
