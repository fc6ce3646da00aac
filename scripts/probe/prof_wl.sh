# rocprofv3 kernel trace + stats of one bench workload, summarised by scripts/prof_summary.py.   usage: prof_wl.sh <workload> <steps> <out-name>
WL=$1; STEPS=$2; OUT=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$OUT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$OUT -o p --output-format csv -- python3 bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline > gpurun_out/prof_$OUT.log 2>&1
python3 scripts/prof_summary.py gpurun_out/prof_$OUT/p gpurun_out/prof_$OUT.log "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline   ($WL, MI355X, round 2)" > gpurun_out/prof_$OUT.txt
rm -rf gpurun_out/prof_$OUT
head -45 gpurun_out/prof_$OUT.txt | cut -c1-150
