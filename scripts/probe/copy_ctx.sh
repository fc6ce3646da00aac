cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cc; mkdir -p gpurun_out/cc
ASM_HIP_TIMING=0 rocprofv3 --kernel-trace -d gpurun_out/cc -o p --output-format csv -- python3 bench.py --workload ${1:-c4} --no-cpu-baseline --steps 10 > gpurun_out/cc/log 2>&1
python3 scripts/probe/copy_ctx.py gpurun_out/cc/p_kernel_trace.csv > gpurun_out/copy_ctx_${1:-c4}.txt
python3 scripts/probe/segments.py gpurun_out/cc/p_kernel_trace.csv > gpurun_out/segments_${1:-c4}.txt
rm -f gpurun_out/cc/p_kernel_trace.csv
tail -1 gpurun_out/cc/log | cut -c1-160
