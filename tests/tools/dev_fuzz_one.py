import sys; sys.path.insert(0, '.')
import numpy as np
from tests.util import random_subproblem, oracle_solve, hip_solve, rel_err
seed = int(sys.argv[1]); n = int(sys.argv[2]); m = int(sys.argv[3]); dens = float(sys.argv[4])
rng = np.random.default_rng(seed)
n2 = int(rng.integers(4, 260)); m2 = int(rng.integers(2, 200))
dens2 = float(rng.choice([1.0, 0.5, 0.1, 0.03])); dup = float(rng.choice([0.0, 0.2])); nr = int(rng.integers(0, min(m2, 6)))
infeas = bool(rng.random() < 0.3); delta = float(rng.choice([0.4, 0.05, 1000.0]))
print('params', n2, m2, dens2, dup, nr, infeas, delta)
sp = random_subproblem(seed, n2, m2, dens2, dup, nr, infeasible=infeas, delta=delta)
qp, o = oracle_solve(sp); opt, h = hip_solve(sp)
print('status', o[5], h[5], o[6]['stats'], opt.last_stats())
rows, bnd, sl = opt.active_set(); orow, obst, osst = o[6]['sets']
print('rows differ at', np.nonzero(rows != orow)[0], 'bounds differ at', np.nonzero(bnd != obst)[0])
for k, nm in enumerate(('p', 'lam', 'mU', 'mL')):
    print(nm, 'rel err', rel_err(h[k], o[k]))
d = np.nonzero(rows != orow)[0]
J = sp['J']
for i in d[:6]:
    if i < m2:
        print('row', i, 'cols', np.nonzero(J[i])[0], 'vals', J[i][np.nonzero(J[i])[0]], 'c_lb', sp['c_lb'][i], 'c_ub', sp['c_ub'][i], 'act', sp['E'][i] + J[i] @ o[0], 'lam o/h', o[1][i], h[1][i])
pobj_o = sp['df'] @ o[0]; pobj_h = sp['df'] @ h[0]
print('objective oracle %.15g hip %.15g' % (pobj_o, pobj_h))
dl = np.nonzero(np.abs(o[1] - h[1]) > 1e-9 * max(1.0, np.abs(o[1]).max()))[0]
print('lam differs at rows', dl)
for i in dl[:10]:
    nzc = np.nonzero(J[i])[0]
    print('  row', i, 'state o/h', orow[i], rows[i], 'cols', nzc, 'vals', np.round(J[i][nzc], 4), 'c_lb', sp['c_lb'][i], 'c_ub', sp['c_ub'][i], 'lam o/h', o[1][i], h[1][i])
cols = sorted(set(int(c) for i in dl for c in np.nonzero(J[i])[0]))
print('columns involved', cols, 'bnd state', obst[cols], 'p', o[0][cols])
for c in cols[:3]:
    rws = np.nonzero(J[:, c])[0]
    print(' col', c, 'active rows on it', [(int(r), int(orow[r]), round(float(o[1][r]), 5), round(float(h[1][r]), 5)) for r in rws if orow[r] or rows[r]])
