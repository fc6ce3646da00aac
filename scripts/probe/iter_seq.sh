cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=$1; MARK=$2
rm -rf gpurun_out/seq; mkdir -p gpurun_out/seq
ASM_HIP_TIMING=0 rocprofv3 --kernel-trace -d gpurun_out/seq -o p --output-format csv -- python3 bench.py --workload $W --no-cpu-baseline --steps 6 > gpurun_out/seq/log 2>&1
python3 scripts/probe/iter_seq.py gpurun_out/seq/p_kernel_trace.csv $MARK > gpurun_out/seq_$W.txt
rm -f gpurun_out/seq/p_kernel_trace.csv
