"""Scratch: which of the first N case300-sized scenarios does not converge, and how its run ends."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf
base = acopf.synthetic_case("case300", 1, 0.5)
names = {0: "warm", 1: "ipm0+ln", 2: "ipm1+ln", 3: "ipm2+ln", 4: "face", 5: "unpol", 6: "infeas", 7: "ph1-infeas", 8: "ipm~", 9: "ref"}
opt = None
for sidx in range(int(sys.argv[1]), int(sys.argv[2])):
    pr = acopf.function_model(acopf.scenario_case(base, sidx)).to_problem("s")
    mdl = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=1000, device_eval=True))
    slp = A.SlpLS(mdl)
    slp.run()
    c = collections.Counter({**names, 10: "conv"}.get(r["stats"]["path"], r["stats"]["path"]) for r in slp.trace)
    if slp.ret != 0:
        last = slp.trace[-1]
        print("scenario", sidx, "status", slp.ret, "LPs", len(slp.trace), dict(c), "last LP status", last["status"], "path", last["stats"]["path"], "fr", last["fr"],
              "prim_infeas %.3e" % slp.prim_infeas, "its", last["stats"]["ipm_iters"])
    slp.optimizer.close()
print("done")
