"""Scratch: step profile of the dense C2 workload (Trust Region)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import problems
pr = problems.synthetic_dense_function_model(1000, 500).to_problem("c2")
mdl = A.Model.from_problem(pr, A.Parameters(algorithm="Trust Region", max_iter=1000, device_eval=True))
slp = A.SlpTR(mdl)
slp.run(max_lp_solves=2)
t0 = time.perf_counter()
slp.run(max_lp_solves=2 + int(sys.argv[1]), resume=True)
print("total %.2f ms/step" % (1e3 * (time.perf_counter() - t0) / int(sys.argv[1])), "LP wall %.2f" % np.mean([r["stats"]["wall_ms"] for r in slp.trace[-int(sys.argv[1]):]]))
