"""Scratch: Line-Search SLP on the case1354pegase-sized synthetic grid at several load scales - which instance leaves
feasibility restoration and terminates?  usage: c4_explore.py <case> <max_lp> <load_scale> [<load_scale> ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf

case_name, max_lp = sys.argv[1], int(sys.argv[2])
names = {0: "warm", 1: "ipm0+ln", 2: "ipm1+ln", 3: "ipm2+ln", 4: "ipm+face", 5: "unpolished", 6: "ipm-infeas", 7: "ph1-infeas", 8: "ipm~+ln", 9: "ipm+ref"}
opt = None
for ls in [float(v) for v in sys.argv[3:]]:
    case = acopf.synthetic_case(case_name, 1, ls)
    pr = acopf.function_model(case).to_problem("%s load %g" % (case_name, ls))
    holder = {}
    def factory(d, r, c):
        holder["opt"] = A.HipSubOptimizer(d, r, c, device=0)
        return holder["opt"]
    mdl = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=1000, external_optimizer=factory, device_eval=True))
    slp = A.SlpLS(mdl)
    t0 = time.perf_counter()
    slp.run(max_lp_solves=max_lp)
    dt = time.perf_counter() - t0
    print("== %s load_scale %g: ret %d after %d LPs, %d iterations, %.1f s (%.0f ms/LP)  inf_pr %.3e" % (case_name, ls, slp.ret, slp.lp_solves, slp.iter, dt, 1e3 * dt / max(slp.lp_solves, 1), slp.prim_infeas), flush=True)
    line = []
    for r in slp.trace:
        st = r["stats"]
        line.append("%s%s/%d" % ("F" if r["fr"] else "N", names.get(st["path"], "?"), st["nfact"]))
    print("   " + " ".join(line), flush=True)
    holder["opt"].close()
