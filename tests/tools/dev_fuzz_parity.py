"""Randomised parity campaign: HIP sub-LP solve vs the oracle on seeded random sub-problems (normal phase, then the
restoration phase on the same handle when the LP is infeasible, then a perturbed warm re-solve)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from tests.util import random_subproblem, oracle_solve, hip_solve, rel_err
N = int(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
PATH = {0: 'warm', 1: 'ipm0+ln', 2: 'ipm1+ln', 3: 'ipm2+ln', 4: 'ipm+ref', 5: 'ipm-unpolished', 6: 'ipm-infeasible', 7: 'phase1-infeasible', 8: 'ipm~+ln'}
bad = 0; paths = {}
def check(tag, o, h, opt):
    global bad
    ok = o[5] == h[5]
    st = opt.last_stats(); so = o[6]['stats']
    ok = ok and PATH[st['path']] == so['path']
    if ok and o[5] == 1 and so['polished'] == 1 and so['path'] != 'ipm+ref':
        rows, bnd, sl = opt.active_set(); orow, obst, osst = o[6]['sets']
        ok = np.array_equal(rows, orow) and np.array_equal(bnd, obst) and np.array_equal(sl, osst)
        errs = [rel_err(h[k], o[k]) for k in range(4)]
        ok = ok and max(errs) < 1e-10
    paths[so['path']] = paths.get(so['path'], 0) + 1
    if not ok:
        bad += 1
        print('MISMATCH', tag, 'status', o[5], h[5], 'paths', so['path'], PATH[st['path']], 'iters', so['ipm_iters'], st['ipm_iters'], flush=True)
t0 = time.time()
for k in range(N):
    rng = np.random.default_rng(seed0 + k)
    n = int(rng.integers(4, 260)); m = int(rng.integers(2, 200))
    dens = float(rng.choice([1.0, 0.5, 0.1, 0.03])); dup = float(rng.choice([0.0, 0.2])); nr = int(rng.integers(0, min(m, 6)))
    infeas = bool(rng.random() < 0.3); delta = float(rng.choice([0.4, 0.05, 1000.0]))
    sp = random_subproblem(seed0 + k, n, m, dens, dup, nr, infeasible=infeas, delta=delta)
    qp, o = oracle_solve(sp); opt, h = hip_solve(sp)
    check('seed %d normal n=%d m=%d dens=%g' % (seed0 + k, n, m, dens), o, h, opt)
    if o[5] == 2:
        qp, o = oracle_solve(sp, True, qp); opt, h = hip_solve(sp, True, opt)
        check('seed %d restoration' % (seed0 + k), o, h, opt)
    sp2 = dict(sp); sp2['dE'] = sp['dE'] * (1.0 + 1e-3 * rng.standard_normal(len(sp['dE'])))
    fr = o[5] == 1 and 'fr' in o[6]['stats'].get('path', '') 
    qp, o = oracle_solve(sp2, False, qp); opt, h = hip_solve(sp2, False, opt)
    check('seed %d perturbed re-solve' % (seed0 + k), o, h, opt)
    opt.close()
print('cases', N, 'mismatches', bad, 'paths', paths, 'time %.1f s' % (time.time() - t0))
