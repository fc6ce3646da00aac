# the committed round-2 C4 profile set: kernel trace + stats of the default bench command, then the PMC passes (scripts/probe/pmc_collect.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c4
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c4 -o p --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/prof_c4.log 2>&1
echo stats rc=$?
python3 scripts/prof_summary.py gpurun_out/prof_c4/p gpurun_out/prof_c4.log "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline   (C4, MI355X, round 2; ASM_HIP_TIMING=1)" > gpurun_out/r02_c4_kernel_stats.txt
rm -rf gpurun_out/prof_c4
bash scripts/probe/pmc_collect.sh
