"""HIP SLP run to termination with a per-LP summary (development probe; run on the GPU box).
usage: fr_trace.py CASE ALGORITHM MAX_ITER [LOAD_SCALE] [-v]"""
import sys, time, collections; sys.path.insert(0, '.')
import activesetmethods_amd as A
from activesetmethods_amd import acopf
name = sys.argv[1]; alg = sys.argv[2]; mi = int(sys.argv[3]); ls = float(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4] != '-v' else 1.0
pr = acopf.acopf_problem(acopf.synthetic_case(name, 1, ls), name)
m = A.Model.from_problem(pr, A.Parameters(algorithm=alg, max_iter=mi))
t0 = time.time()
s = A.optimize(m)
dt = time.time() - t0
paths = collections.Counter((r['stats']['path'], r['fr']) for r in s.trace)
bad = [(k, r['status'], r['stats']['path'], r['stats']['polished']) for k, r in enumerate(s.trace) if r['status'] not in (1, 2) or r['stats']['polished'] != 1]
print(name, alg, 'load', ls, 'status', m.status, 'iter', s.iter, 'lp', s.lp_solves, 'time %.2f s' % dt, 'inf_pr %.2e' % s.prim_infeas,
      'paths (path, fr):count', dict(paths), 'bad', bad)
if '-v' in sys.argv:
    for k, r in enumerate(s.trace):
        st = r['stats']
        print(k, 'fr', r['fr'], 'status', r['status'], 'path', st['path'], 'ipm', st['ipm_iters'], 'eqp', st['eqp'], 'pol', st['polished'], 'ms %.1f' % st['wall_ms'])
