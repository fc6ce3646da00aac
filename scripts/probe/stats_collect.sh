# Kernel-trace summary of one workload for profiles/ (no PMC passes): bash scripts/probe/stats_collect.sh <round tag> <workload> [extra bench args]
TAG=${1:-r03}; WL=${2:-c3}; shift; shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd $ROOT
OUT=gpurun_out/stats_${TAG}_${WL}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o p --output-format csv -- python3 bench.py --workload $WL --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
echo stats rc=$?
python3 scripts/prof_summary.py $OUT/stats/p $OUT/stats.log "rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $WL --no-cpu-baseline $*   (MI355X, round ${TAG#r}; ASM_HIP_TIMING=1)" > gpurun_out/${TAG}_${WL}_kernel_stats.txt
rm -f $OUT/stats/p_kernel_trace.csv
