for cfg in "1,1" "1,4" "2,4" "2,8" "1,8"; do
  echo "ASM_NS_SPLIT=$cfg"
  ASM_NS_SPLIT=$cfg ASM_HIP_VERBOSE=1 python scripts/probe/step_profile.py case1354pegase 0.5 6 2>&1 | grep "phases\|total" | tail -4
done
