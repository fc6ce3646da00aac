#!/usr/bin/env python3
"""<tag>_<wl>_pmc_mfma.txt, <tag>_<wl>_pmc_traffic.{txt,json} from the per-kernel sums scripts/probe/pmc_collect.sh
leaves under gpurun_out/pmc_<tag>_<wl>/{mfma,clock,fetch,write}/ (three separate rocprofv3 --pmc passes of the same command).
FETCH_SIZE is doubled for gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes (wide coalesced reads are tallied at 1/2).
usage: pmc_summaries.py <round tag> <workload> "<profiled command>"   (writes gpurun_out/<tag>_<wl>_pmc_*.{txt,json}; copy to profiles/)"""
import json, sys
tag, wl, CMD = sys.argv[1], sys.argv[2], sys.argv[3]
base = "gpurun_out/pmc_%s_%s" % (tag, wl)
M = json.load(open(base + "/mfma/m_summary.json"))["sums"]
F = json.load(open(base + "/fetch/f_summary.json"))["sums"]
W = json.load(open(base + "/write/w_summary.json"))["sums"]
try:
    CK = json.load(open(base + "/clock/c_summary.json"))["sums"]
except Exception:
    CK = {}
with open("gpurun_out/%s_%s_pmc_mfma.txt" % (tag, wl), "w") as f:
    f.write("# rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace -- %s\n" % CMD)
    f.write("# (workload %s, MI355X, round %s; counters collected in their own pass: kernels run serialised under --pmc, so these are the kernels ALONE on the chip)\n" % (wl, tag[1:]))
    f.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 2.4 GHz x 1024 SIMDs); tflops = SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 / kernel time;\n")
    f.write("# wait_any / wait_inst / active / wait_lds = share of SQ_WAVE_CYCLES\n")
    f.write("# clock_GHz = GRBM_GUI_ACTIVE / 8 / kernel time (own pass; MI355X_MICROARCH.md, DVFS give-back); busy@clk = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs):\n")
    f.write("# the share of the cycles the chip actually ran; waves/SIMD = SQ_WAVE_CYCLES x 4 / (GRBM_GUI_ACTIVE / 8 x 1024)\n")
    f.write("%-28s %8s %10s %9s %8s %9s %9s %8s %9s %10s %9s %11s\n" % ("kernel", "launches", "time_ms", "mfma_util", "tflops", "wait_any", "wait_inst", "active", "wait_lds", "clock_GHz", "busy@clk", "waves/SIMD"))
    for k, v in sorted(M.items(), key=lambda kv: -kv[1].get("_duration_ns", 0))[:16]:
        d = v.get("_duration_ns", 0.0) or 1.0
        wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        ck = CK.get(k, {})
        cyc = ck.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        cd = ck.get("_duration_ns", 0.0) or 1.0
        f.write("%-28s %8d %10.1f %9.3f %8.2f %9.3f %9.3f %8.3f %9.3f %10.3f %9.3f %11.2f\n" % (k[:28], v.get("_launches", 0), d / 1e6, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d * 2.4 * 1024),
                v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512 / d / 1e3, v.get("SQ_WAIT_ANY", 0.0) / wc, v.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, v.get("SQ_WAIT_INST_LDS", 0.0) / wc, cyc / cd,
                ck.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0) if cyc else 0.0, ck.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / (cyc * 1024.0) if cyc else 0.0))
syrk = sorted(k for k in F if k.startswith("k_syrk"))
per = {}
for k in syrk:
    n = F[k].get("_launches", 0) or 1
    per[k] = dict(launches=n, fetch_bytes=2.0 * F[k].get("FETCH_SIZE", 0.0) * 1024.0, write_bytes=W.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0,
                  time_ms=F[k].get("_duration_ns", 0.0) / 1e6)
L = sum(v["launches"] for v in per.values())
doc = {"_comment": "HBM traffic per launch of the dominant kernel family (k_syrk_upd: Cholesky updates; k_syrk<T,NW,KC,WPE>: Schur builds and small updates) on workload %s: rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in separate passes "
                   "(KB; FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes), command: %s.  Made by scripts/pmc_summaries.py." % (wl, CMD),
       "workload": wl, "kernel": "k_syrk_upd + k_syrk<T,NW,KC,WPE> (all rank-K launches)", "launches": L,
       "fetch_bytes_per_launch": sum(v["fetch_bytes"] for v in per.values()) / L, "write_bytes_per_launch": sum(v["write_bytes"] for v in per.values()) / L,
       "traffic_bytes_per_launch": sum(v["fetch_bytes"] + v["write_bytes"] for v in per.values()) / L, "per_instantiation": per}
# every kernel (bench.py reads the entry of the kernel family it reports as dominant; the rank-K family without k_syrk_upd where that
# kernel only occurs in the cold first LP of the run, which the timed region does not contain)
allk = {}
for k in F:
    n = F[k].get("_launches", 0) or 1
    allk[k] = dict(launches=n, fetch_bytes_per_launch=2.0 * F[k].get("FETCH_SIZE", 0.0) * 1024.0 / n, write_bytes_per_launch=W.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0 / n,
                   ms_per_launch=F[k].get("_duration_ns", 0.0) / 1e6 / n)
doc["per_kernel"] = allk
def fam(names):
    ks = [k for k in allk if any(k.startswith(nm) for nm in names)]
    n = sum(allk[k]["launches"] for k in ks) or 1
    return dict(kernels=ks, launches=n, traffic_bytes_per_launch=sum((allk[k]["fetch_bytes_per_launch"] + allk[k]["write_bytes_per_launch"]) * allk[k]["launches"] for k in ks) / n)
doc["families"] = {"panel_kernel": fam(["k_chol_panel"]), "syrk_kernel": fam(["k_syrk"]), "syrk_kernel_without_upd": fam(["k_syrk<"])}
json.dump(doc, open("gpurun_out/%s_%s_pmc_traffic.json" % (tag, wl), "w"), indent=1)
with open("gpurun_out/%s_%s_pmc_traffic.txt" % (tag, wl), "w") as f:
    f.write("# HBM traffic of the k_syrk kernels (PMC, separate passes, one counter per pass), MI355X, round %s\n" % tag[1:])
    f.write("#   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- %s\n#   rocprofv3 --pmc WRITE_SIZE --kernel-trace -- %s\n" % (CMD, CMD))
    f.write("# FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE x2 (gfx950: wide coalesced reads are tallied at 1/2, MI355X_MICROARCH.md, HBM section).\n")
    f.write("%-24s %8s %16s %16s %12s\n" % ("kernel", "launches", "fetch MB/launch", "write MB/launch", "ms/launch"))
    for k, v in per.items():
        f.write("%-24s %8d %16.1f %16.1f %12.3f\n" % (k, v["launches"], v["fetch_bytes"] / v["launches"] / 1e6, v["write_bytes"] / v["launches"] / 1e6, v["time_ms"] / v["launches"]))
    g = F.get("k_gemv_n", None)
print(open("gpurun_out/%s_%s_pmc_mfma.txt" % (tag, wl)).read())
print(open("gpurun_out/%s_%s_pmc_traffic.txt" % (tag, wl)).read())
