#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters of a rocprofv3 --pmc pass and the matrix-core utilisation derived from them.
usage: pmc_mfma.py <results.db> > profiles/<name>.txt
  MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (per-SIMD busy cycles over the cycles the SQ had work);
  f64 matrix flops   = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 (the counter counts units of 512 flops, MI355X_MICROARCH.md)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
dur = dict(c.execute("select name, sum(end-start) from kernels group by name").fetchall())
K = {}
for name, ctr, nd, v in rows:
    K.setdefault(name, {})[ctr] = v
    K[name]["_launches"] = nd
names = sorted(K, key=lambda n: -K[n].get("SQ_BUSY_CYCLES", 0.0))
ctrs = ["SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"]
print("# per-kernel sums over all dispatches; mfma_busy = MFMA_BUSY/BUSY ; wait_any, wait_inst, active = share of WAVE_CYCLES")
print("%-40s %8s %10s %9s %9s %9s %9s %9s %12s" % ("kernel", "launches", "time_ms", "mfma_busy", "wait_any", "wait_inst", "active", "wait_lds", "mops_f64"))
for n in names[:14]:
    d = K[n]
    busy = d.get("SQ_BUSY_CYCLES", 0.0) or 1.0
    wc = d.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    print("%-40s %8d %10.1f %9.3f %9.3f %9.3f %9.3f %9.3f %12.4g" % (n.split("(")[0].replace("void ", "")[:40], d["_launches"], dur.get(n, 0) / 1e6,
          d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / busy, d.get("SQ_WAIT_ANY", 0.0) / wc, d.get("SQ_WAIT_INST_ANY", 0.0) / wc,
          d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, d.get("SQ_WAIT_INST_LDS", 0.0) / wc, d.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)))
print("# raw sums")
for n in names[:6]:
    print(n.split("(")[0].replace("void ", ""), {k: K[n].get(k) for k in ctrs})
