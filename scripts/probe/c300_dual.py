"""case300-sized restoration LP: HIP vs oracle, where do the multipliers differ (development probe; GPU box)."""
import sys; sys.path.insert(0, '.')
import numpy as np
from activesetmethods_amd import acopf
from tests.util import oracle_solve, hip_solve, rel_err
pr = acopf.acopf_problem(acopf.synthetic_case("case300", 1, 0.5), "case300")
x = pr.x0.copy()
sp = dict(n=pr.n, m=pr.m, j_row=pr.j_row, j_col=pr.j_col, dE=pr.eval_jac_g(x, np.zeros(pr.nnz)), df=pr.eval_grad_f(x, np.zeros(pr.n)),
          f=pr.eval_f(x), E=pr.eval_g(x, np.zeros(pr.m)), x_k=x, c_lb=pr.g_L, c_ub=pr.g_U, v_lb=pr.x_L, v_ub=pr.x_U, delta=0.05)
qp, o = oracle_solve(sp); opt, h = hip_solve(sp)
qp, o = oracle_solve(sp, True, qp); opt, h = hip_solve(sp, True, opt)
print('stats hip', opt.last_stats()); print('stats oracle', o[6]['stats'])
d = np.abs(h[1] - o[1]); idx = np.argsort(-d)[:10]
print('lam diffs', [(int(i), float(h[1][i]), float(o[1][i])) for i in idx])
print('count > 1e-9', int((d > 1e-9).sum()), 'of', len(d))
for k in range(4): print(k, rel_err(h[k], o[k]))
