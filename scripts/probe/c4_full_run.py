import os, sys
sys.path.insert(0, os.getcwd())
import activesetmethods_amd as A
from activesetmethods_amd import acopf
pr = acopf.acopf_problem(acopf.synthetic_case("case1354pegase", 1, 0.5), "c4")
m = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=100))
s = A.optimize(m)
for k, r in enumerate(s.trace):
    print(k, r['status'], r['stats']['path'], r['stats']['polished'], r['stats'].get('ipm_iters'), file=sys.stdout)
print("status", m.status, s.iter)
