"""Is the nominal-load synthetic case1354pegase-sized instance feasible?    Line-Search SLP runs at several load
scales: status, iterations, restoration LPs and the primal infeasibility the run ends with; plus the aggregate balance of the grid
(total load against total generation capacity), which bounds feasibility from above whatever the network does."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf

case = sys.argv[1] if len(sys.argv) > 1 else "case1354pegase"
scales = [float(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0.5", "0.7", "0.8", "0.9", "1.0"])]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 80
for sc in scales:
    c = acopf.synthetic_case(case, 1, sc)
    pd, qd = float(np.sum(c["pd"])), float(np.sum(c["qd"]))
    pmax, qmax = float(np.sum(c["pmax"])), float(np.sum(c["qmax"]))
    pr = acopf.function_model(c).to_problem("%s load %g" % (case, sc))
    t0 = time.perf_counter()
    slp = A.optimize(A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=iters, device_eval=True)))
    dt = time.perf_counter() - t0
    fr = sum(1 for r in slp.trace if r["fr"])
    tail = [round(float(np.max(np.abs(r["p"]))), 6) for r in slp.trace[-3:]]
    print("load scale %.2f: sum pd %.1f of sum pmax %.1f (%.0f %%), sum qd %.1f of sum qmax %.1f | status %d after %d LPs (%d restoration), "
          "final inf_pr %.3e inf_du %.3e, last steps |p|_inf %s, %.1f s" % (sc, pd, pmax, 100 * pd / pmax, qd, qmax, slp.ret, len(slp.trace), fr, slp.prim_infeas,
                                                                          slp.dual_infeas, tail, dt), flush=True)
    slp.optimizer.close()
