"""Scratch: host time of the pieces of SlpLS.sub_optimize around the library call (WITH_TORCH=1 imports torch first, as bench.py does)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if os.environ.get('WITH_TORCH'):
    import torch
    torch.cuda.synchronize()
import activesetmethods_amd as A
from activesetmethods_amd import acopf, slp as S, subproblem as SP
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
for m in ("solve_resident", "last_stats", "active_set"):
    wrap(SP.HipSubOptimizer, m)
wrap(S.SlpLS, "sub_optimize")
mdl = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=1000, device_eval=True))
slp = A.SlpLS(mdl)
slp.run(max_lp_solves=2)
opt = slp.optimizer
raw = opt._lib.asm_sublp_solve_resident
class Timed:
    def __init__(self, f): self.f = f
    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.f(*a)
        T["C call"] = T.get("C call", 0.0) + time.perf_counter() - t0
        return r
class LibProxy:
    def __init__(self, lib): self._lib = lib; self.asm_sublp_solve_resident = Timed(lib.asm_sublp_solve_resident)
    def __getattr__(self, k): return getattr(self._lib, k)
opt._lib = LibProxy(opt._lib)
T.clear()
slp.run(max_lp_solves=2 + nlp, resume=True)
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-22s %.2f ms/step" % (k, 1e3 * v / nlp))
print("  LP wall inside library %.2f ms" % np.mean([r["stats"]["wall_ms"] for r in slp.trace[-nlp:]]))
