"""Scratch: the 8 case300-sized scenarios of the batch test one by one: status, iterations, final infeasibility."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf
base = acopf.synthetic_case("case300", 1, 0.5)
maxit = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for sidx in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    pr = acopf.acopf_problem(acopf.scenario_case(base, sidx), "s%d" % sidx)
    m = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=maxit))
    t0 = time.time()
    s = A.optimize(m)
    tr = s.trace
    print(sidx, "ret", s.ret, "iters", s.iter, "lps", s.lp_solves, "inf_pr %.3e inf_du %.3e compl %.3e |p| %.2e alpha %.2e" % (s.prim_infeas, s.dual_infeas, s.compl, np.abs(s.p).max(), s.alpha),
          "fr", sum(1 for r in tr if r["fr"]), "paths", sorted(set(r["stats"]["path"] for r in tr)), "%.1fs" % (time.time() - t0), flush=True)
    s.optimizer.close()
