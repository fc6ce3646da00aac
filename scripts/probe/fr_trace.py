"""HIP SLP run with the solver's verbose log (development probe; run on the GPU box)."""
import sys; sys.path.insert(0, '.')
import activesetmethods_amd as A
from activesetmethods_amd import acopf
name = sys.argv[1]; alg = sys.argv[2]; mi = int(sys.argv[3]); ls = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
pr = acopf.acopf_problem(acopf.synthetic_case(name, 1, ls), name)
m = A.Model.from_problem(pr, A.Parameters(algorithm=alg, max_iter=mi))
s = A.optimize(m)
print('status', m.status, 'iter', s.iter, 'lp', s.lp_solves)
for k, r in enumerate(s.trace):
    st = r['stats']
    print(k, 'fr', r['fr'], 'status', r['status'], 'path', st['path'], 'ipm', st['ipm_iters'], 'eqp', st['eqp'], 'pol', st['polished'], 'ms %.1f' % st['wall_ms'])
