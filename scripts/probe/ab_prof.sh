cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in libasmhip_base.so libasmhip.so; do
  export ASM_HIP_LIB=$lib
  rm -rf gpurun_out/abp_$lib; mkdir -p gpurun_out/abp_$lib
  rocprofv3 --kernel-trace --stats -d gpurun_out/abp_$lib -o p --output-format csv -- python3 bench.py --workload c4 --no-cpu-baseline > gpurun_out/abp_$lib/log 2>&1
  python3 scripts/prof_summary.py gpurun_out/abp_$lib/p gpurun_out/abp_$lib/log "$lib" | head -12 | cut -c1-40,100-150
  rm -f gpurun_out/abp_$lib/p_kernel_trace.csv
done
