"""Per SLP iteration: LP path / factorisations and the distance of the optimal active set from the previous LP's."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf, problems
name = sys.argv[1]; nlp = int(sys.argv[2]); alg = sys.argv[3] if len(sys.argv) > 3 else "Line Search"
pr = acopf.acopf_problem(acopf.synthetic_case(name, 1), name) if name.startswith("case") else problems.synthetic_dense_nlp()
m = A.Model.from_problem(pr, A.Parameters(algorithm=alg, max_iter=10**6))
slp = A.SlpLS(m) if alg == "Line Search" else A.SlpTR(m)
prev = None
for it in range(nlp):
    t = time.time(); slp.run(max_lp_solves=it + 1, resume=it > 0); t = time.time() - t
    st = slp.trace[-1]['stats']; act = slp.optimizer.active_set()
    msg = ""
    if act is not None:
        rows, bnd, sl = act
        if prev is not None and len(prev[0]) == len(rows) and slp.trace[-1]['fr'] == prevfr:
            dr = int((rows != prev[0]).sum()); db = int((bnd != prev[1]).sum())
            add = int(((rows == 1) & (prev[0] == 0)).sum()); drop = int(((rows == 0) & (prev[0] == 1)).sum())
            msg = "rows changed %d (+%d -%d) of %d active; bounds changed %d of %d active" % (dr, add, drop, int(rows.sum()), db, int((bnd != 0).sum()))
        prev = (rows.copy(), bnd.copy()); prevfr = slp.trace[-1]['fr']
    print("it %2d %.2fs fr %d status %d path %d ipm %d nfact %d eqp %d | %s | infeas %.2e" % (it, t, slp.trace[-1]['fr'], slp.trace[-1]['status'], st['path'], st['ipm_iters'], st['nfact'], st['eqp'], msg, slp.prim_infeas), flush=True)
