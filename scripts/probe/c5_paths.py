"""Scratch: path histogram, per-LP time and set changes along one complete case300-sized scenario solve."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf
base = acopf.synthetic_case("case300", 1, 0.5)
pr = acopf.function_model(acopf.scenario_case(base, int(sys.argv[1]))).to_problem("s")
mdl = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=1000, device_eval=True))
slp = A.SlpLS(mdl)
slp.run()
names = {0: "warm", 1: "ipm0+ln", 2: "ipm1+ln", 3: "ipm2+ln", 4: "face", 5: "unpol", 9: "ref"}
print("status", slp.ret, "LPs", len(slp.trace), collections.Counter(names.get(r["stats"]["path"], r["stats"]["path"]) for r in slp.trace))
prev = None
for k, r in enumerate(slp.trace):
    s = r.get("sets")
    chg = -1
    if s is not None and prev is not None:
        chg = int((s[0] != prev[0]).sum() + (s[1] != prev[1]).sum())
    prev = s
    print(k, names.get(r["stats"]["path"]), "its", r["stats"]["ipm_iters"], "ms %.2f" % r["stats"]["wall_ms"], "set changes", chg, "alpha %.3g" % 0)
