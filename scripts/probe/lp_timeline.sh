cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
ASM_HIP_TIMING=0 rocprofv3 --kernel-trace -d gpurun_out/tl -o p --output-format csv -- python3 bench.py --workload c4 --no-cpu-baseline --steps 10 > gpurun_out/tl/log 2>&1
python3 scripts/probe/lp_timeline.py gpurun_out/tl/p_kernel_trace.csv 6 > gpurun_out/tl_c4.txt
rm -f gpurun_out/tl/p_kernel_trace.csv
tail -1 gpurun_out/tl/log | cut -c1-160
