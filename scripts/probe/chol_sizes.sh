# panel-kernel duration per matrix size (one profiled run per size): bash scripts/probe/chol_sizes.sh N ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in "$@"; do
  rm -rf gpurun_out/cs; mkdir -p gpurun_out/cs
  rocprofv3 --kernel-trace --stats -d gpurun_out/cs -o p --output-format csv -- python3 scripts/probe/chol_time.py $n > gpurun_out/cs/log 2>&1
  python3 - gpurun_out/cs/p_kernel_stats.csv $n <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'chol_panel' in r['Name']: print("N %-12s %-22s calls %4s avg %8.1f us  min %8.1f" % (sys.argv[2], r['Name'].split('(')[0][:22], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3), flush=True)
PY
done
rm -rf gpurun_out/cs
