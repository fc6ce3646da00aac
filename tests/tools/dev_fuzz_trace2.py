import sys; sys.path.insert(0, '.')
import numpy as np
from tests.util import random_subproblem, oracle_solve, hip_solve
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
n = int(rng.integers(4, 260)); m = int(rng.integers(2, 200))
dens = float(rng.choice([1.0, 0.5, 0.1, 0.03])); dup = float(rng.choice([0.0, 0.2])); nr = int(rng.integers(0, min(m, 6)))
infeas = bool(rng.random() < 0.3); delta = float(rng.choice([0.4, 0.05, 1000.0]))
print('params n m dens dup nr infeas delta', n, m, dens, dup, nr, infeas, delta)
sp = random_subproblem(seed, n, m, dens, dup, nr, infeasible=infeas, delta=delta)
qp, o = oracle_solve(sp); opt, h = hip_solve(sp)
print('first', o[5], h[5], o[6]['stats']['path'], opt.last_stats()['path'])
if o[5] == 2:
    qp, o = oracle_solve(sp, True, qp); opt, h = hip_solve(sp, True, opt)
    print('FR', o[5], h[5], o[6]['stats']['path'], opt.last_stats()['path'])
sp2 = dict(sp); sp2['dE'] = sp['dE'] * (1.0 + 1e-3 * rng.standard_normal(len(sp['dE'])))
from oracle import lp_solver as L
orig_run = L.IPM.run
def run(self, tol, more):
    st = orig_run(self, tol, more)
    for e in self.log[-8:]: print('  oracle ipm', e[0], '%.3e %.3e %.3e' % e[1:], 'ymax %.3e' % np.abs(self.y).max())
    print('  farkas', L.farkas_margin(self.lp, self.y))
    return st
L.IPM.run = run
qp, o = oracle_solve(sp2, False, qp); print('oracle re-solve', o[5], o[6]['stats'])
opt, h = hip_solve(sp2, False, opt); print('hip re-solve', h[5], opt.last_stats())
