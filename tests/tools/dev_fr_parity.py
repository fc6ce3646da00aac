"""HIP vs oracle on an SLP run that enters the restoration phase (non-unique LP optima, path 'ipm+ref')."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf
from oracle import slp as O
name = sys.argv[1]; nlp = int(sys.argv[2]); alg = sys.argv[3] if len(sys.argv) > 3 else "Trust Region"
PATH_NAMES = {0: 'warm', 1: 'ipm0+ln', 2: 'ipm1+ln', 3: 'ipm2+ln', 4: 'ipm+ref', 5: 'ipm-unpolished', 6: 'ipm-infeasible', 7: 'phase1-infeasible', 8: 'ipm~+ln'}
pr = acopf.acopf_problem(acopf.synthetic_case(name, 1), name)
mh = A.Model.from_problem(pr, A.Parameters(algorithm=alg, max_iter=nlp))
sh = A.optimize(mh)
mo = O.Model(pr.n, pr.m, pr.x_L, pr.x_U, pr.g_L, pr.g_U, pr.j_str, pr.eval_f, pr.eval_g, pr.eval_grad_f, pr.eval_jac_g, O.Parameters(algorithm=alg, max_iter=nlp))
mo.x[:] = pr.x0
so = O.optimize(mo)
for k, (a, b) in enumerate(zip(sh.trace, so.trace)):
    sa, sb = a['stats'], b['stats']
    dp = np.abs(np.asarray(a['p']) - np.asarray(b['p'])).max() / max(1.0, np.abs(np.asarray(b['p'])).max())
    dl = np.abs(np.asarray(a['lam']) - np.asarray(b['lam'])).max() / max(1.0, np.abs(np.asarray(b['lam'])).max())
    print(k, 'fr', a['fr'], b['fr'], 'status', a['status'], b['status'], PATH_NAMES[sa['path']], sb['path'], 'ipm', sa['ipm_iters'], sb['ipm_iters'], 'eqp', sa['eqp'], sb['eqp'], 'dp %.1e dlam %.1e' % (dp, dl))
print('x diff', np.abs(mh.x - mo.x).max())
