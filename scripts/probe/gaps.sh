# idle gaps of the GPU inside a bench run, attributed to the kernel that ran BEFORE the gap.   usage: gaps.sh <workload> <steps>
WL=$1; STEPS=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gaps_$WL
rocprofv3 --kernel-trace -d gpurun_out/gaps_$WL -o p --output-format csv -- python3 bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline > gpurun_out/gaps_$WL.log 2>&1
python3 - $WL <<'PY'
import csv, glob, sys, collections
wl = sys.argv[1]
rows = []
for fn in glob.glob("gpurun_out/gaps_%s/*kernel_trace.csv" % wl):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
# skip the set-up phase: start at the first k_ipm_measures
i0 = next(i for i, r in enumerate(rows) if r[2].startswith("k_ipm_measures"))
rows = rows[i0:]
gap = collections.defaultdict(float); cnt = collections.Counter()
end = rows[0][1]; busy = 0.0; prev = rows[0][2]
for s, e, n in rows[1:]:
    if s > end:
        g = s - end
        gap[prev] += g; cnt[prev] += 1
    if e > end:
        busy += e - max(s, end); end = e; prev = n
span = rows[-1][1] - rows[0][0]
tot = sum(gap.values())
with open("gpurun_out/gaps_%s.txt" % wl, "w") as f:
    f.write("# span %.1f ms, idle %.1f ms (%.1f %%), by the kernel that ran before the gap\n" % (span / 1e6, tot / 1e6, 100.0 * tot / span))
    f.write("%-34s %8s %10s %10s\n" % ("after kernel", "gaps", "idle_ms", "avg_us"))
    for k, v in sorted(gap.items(), key=lambda kv: -kv[1])[:22]:
        f.write("%-34s %8d %10.2f %10.2f\n" % (k[:34], cnt[k], v / 1e6, v / cnt[k] / 1e3))
print(open("gpurun_out/gaps_%s.txt" % wl).read())
PY
rm -rf gpurun_out/gaps_$WL
