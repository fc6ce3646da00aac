"""Scratch: where one SLP step of a workload spends its host time (evaluation, reductions, LP call, merit line search)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if os.environ.get('WITH_TORCH'):
    import torch
    torch.cuda.synchronize()
import activesetmethods_amd as A
from activesetmethods_amd import acopf, slp as S
name, ls, nlp = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
case = acopf.synthetic_case(name, 1, ls)
pr = acopf.function_model(case).to_problem(name)
T = {}
def wrap(cls, meth):
    f = getattr(cls, meth)
    def g(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return f(self, *a, **k)
        finally:
            T[meth] = T.get(meth, 0.0) + time.perf_counter() - t0
    setattr(cls, meth, g)
for m in ("eval_functions", "norm_violations", "KT_residuals", "norm_complementarity", "sub_optimize", "compute_nu", "compute_phi", "compute_derivative", "compute_alpha"):
    wrap(S.SlpLS, m)
mdl = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=1000, device_eval=True))
slp = A.SlpLS(mdl)
slp.run(max_lp_solves=int(sys.argv[4]) if len(sys.argv) > 4 else 3)
T.clear()
t0 = time.perf_counter()
slp.run(max_lp_solves=(int(sys.argv[4]) if len(sys.argv) > 4 else 3) + nlp, resume=True)
tot = time.perf_counter() - t0
print("total %.1f ms/step" % (1e3 * tot / nlp))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-22s %.2f ms/step" % (k, 1e3 * v / nlp))
lpw = np.mean([r["stats"]["wall_ms"] for r in slp.trace[-nlp:]])
print("  LP wall inside library %.2f ms" % lpw)
