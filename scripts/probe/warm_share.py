"""Path histogram of complete SLP runs per third of the run: how often does the retained working set
(path 0 'warm') answer an LP, how many factorisations / interior-point iterations does an LP cost early and late?
   python scripts/probe/warm_share.py [case1354pegase:0.5 case300:0.5 case118:1.0 ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf

names = {0: "warm", 1: "ipm0+ln", 2: "ipm1+ln", 3: "ipm2+ln", 4: "ipm+face", 5: "unpolished", 6: "infeasible", 7: "phase1-inf", 8: "ipm~+ln", 9: "ipm+ref", 10: "ipm-conv"}
for spec in (sys.argv[1:] or ["case1354pegase:0.5", "case300:0.5", "case118:1.0"]):
    case, load = spec.split(":")
    pr = acopf.function_model(acopf.synthetic_case(case, 1, float(load))).to_problem(spec)
    t0 = time.perf_counter()
    slp = A.optimize(A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=300, device_eval=True)))
    dt = time.perf_counter() - t0
    tr = slp.trace
    print("%s: status %d after %d LPs, %.2f s" % (spec, slp.ret, len(tr), dt))
    for k in range(3):
        part = tr[len(tr) * k // 3: len(tr) * (k + 1) // 3]
        hist = {}
        for r in part:
            hist[names[r["stats"]["path"]]] = hist.get(names[r["stats"]["path"]], 0) + 1
        chg = []
        for a, b in zip(part[:-1], part[1:]):
            if a.get("sets") is not None and b.get("sets") is not None:
                chg.append(int(sum(np.count_nonzero(x != y) for x, y in zip(a["sets"], b["sets"]))))
        print("  third %d: %3d LPs  paths %s  factorisations/LP %.1f  ipm iterations/LP %.1f  LP ms %.1f  working-set changes between consecutive LPs: median %s max %s"
              % (k + 1, len(part), hist, np.mean([r["stats"]["nfact"] for r in part]), np.mean([r["stats"]["ipm_iters"] for r in part]),
                 np.mean([r["stats"]["wall_ms"] for r in part]), int(np.median(chg)) if chg else None, max(chg) if chg else None))
    slp.optimizer.close()
