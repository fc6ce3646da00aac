# per-kernel durations of the Cholesky test hook for two builds in one GPU call: bash scripts/probe/chol_prof.sh N[:band] ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in ${LIBS:-libasmhip_base.so libasmhip.so}; do
  export ASM_LIB=$lib
  rm -rf gpurun_out/cp_$lib; mkdir -p gpurun_out/cp_$lib
  rocprofv3 --kernel-trace --stats -d gpurun_out/cp_$lib -o p --output-format csv -- python3 scripts/probe/chol_time.py "$@" > gpurun_out/cp_$lib/log 2>&1
  echo "# $lib"; grep "^N" gpurun_out/cp_$lib/log
  python3 - gpurun_out/cp_$lib/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'chol_panel' in r['Name'] or 'trtri' in r['Name']: print("  %-28s calls %5s avg %9.1f us" % (r['Name'].split('(')[0][:28], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -f gpurun_out/cp_$lib/p_kernel_trace.csv
done
