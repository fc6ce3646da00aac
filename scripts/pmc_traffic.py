#!/usr/bin/env python3
"""HBM traffic per launch of the k_syrk kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
separate runs, values in KB).  FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
(calibration: the dense 1.67 GB gemv operand used to read 816 MB x 2).  Writes the JSON that bench.py attaches to
`roofline.traffic`.   usage: pmc_traffic.py <fetch.db> <write.db> <workload> > profiles/r01_<workload>_pmc_traffic.json"""
import json, sqlite3, sys
fdb, wdb, wl = sys.argv[1:4]
def per_kernel(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, cnt, tot in c.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (counter,)):
        out[name.split("(")[0].replace("void ", "")] = (cnt, tot * 1024.0)
    return out
F = per_kernel(fdb, "FETCH_SIZE"); W = per_kernel(wdb, "WRITE_SIZE")
syrk = sorted(k for k in F if k.startswith("k_syrk"))
launches = sum(F[k][0] for k in syrk)
fetch = 2.0 * sum(F[k][1] for k in syrk)
write = sum(W[k][1] for k in syrk if k in W)
doc = {
    "_comment": "HBM traffic per launch of the dominant kernel family k_syrk<T,NW,KC,WPE> on workload %s (one sub-LP), rocprofv3 PMC passes "
                "(FETCH_SIZE and WRITE_SIZE in separate runs, KB; FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes). "
                "Made by scripts/pmc_traffic.py; details: profiles/r01_c4_pmc_traffic.txt" % wl,
    "workload": wl, "kernel": "k_syrk<T,NW,KC,WPE> (all instantiations)", "launches": launches,
    "fetch_bytes_per_launch": fetch / launches, "write_bytes_per_launch": write / launches,
    "traffic_bytes_per_launch": (fetch + write) / launches,
    "per_instantiation": {k: {"launches": F[k][0], "fetch_bytes": 2.0 * F[k][1], "write_bytes": W.get(k, (0, 0.0))[1]} for k in syrk},
}
print(json.dumps(doc, indent=1))
