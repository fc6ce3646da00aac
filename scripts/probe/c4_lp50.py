"""Scratch: the C4 termination run up to a given LP (ASM_HIP_VERBOSE=1 for the library's log); prints the LPs that did not end on a least-norm polish."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import activesetmethods_amd as A
from activesetmethods_amd import acopf
pr = acopf.acopf_problem(acopf.synthetic_case("case1354pegase", 1, 0.5), "c")
mdl = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=100))
slp = A.SlpLS(mdl)
slp.run(max_lp_solves=int(sys.argv[1]))
print([(k, r['status'], r['stats']['path'], r['stats']['ipm_iters']) for k, r in enumerate(slp.trace) if r['stats']['path'] not in (1, 2)])
