cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c5
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c5 -o p --output-format csv -- python3 bench.py --workload c5 --gpus 1 --steps 8 --warmup 0 --concurrency 4 --no-cpu-baseline > gpurun_out/prof_c5.log 2>&1
python3 scripts/prof_summary.py gpurun_out/prof_c5/p gpurun_out/prof_c5.log "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload c5 --gpus 1 --steps 8 --warmup 0 --concurrency 4" > gpurun_out/prof_c5.txt
rm -rf gpurun_out/prof_c5
head -30 gpurun_out/prof_c5.txt | cut -c1-150
