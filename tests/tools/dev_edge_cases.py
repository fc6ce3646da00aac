"""Edge cases of the sub-LP boundary: no rows, no Jacobian entries, one variable, fixed variables."""
import sys; sys.path.insert(0, '.')
import numpy as np
from tests.util import oracle_solve, hip_solve, rel_err
INF = np.inf
def case(name, sp):
    try:
        qp, o = oracle_solve(sp)
        print(name, 'oracle status', o[5], 'p', np.round(o[0], 6), 'lam', np.round(o[1], 6))
    except Exception as e:
        print(name, 'oracle EXC', repr(e)); o = None
    try:
        opt, h = hip_solve(sp)
        print(name, 'hip    status', h[5], 'p', np.round(h[0], 6), 'lam', np.round(h[1], 6), opt.last_stats()['path'])
        if o is not None and o[5] == h[5] == 1:
            print('   rel err p', rel_err(h[0], o[0]), 'lam', rel_err(h[1], o[1]), 'mU', rel_err(h[2], o[2]), 'mL', rel_err(h[3], o[3]))
        opt.close()
    except Exception as e:
        print(name, 'hip EXC', repr(e))
e = np.zeros(0)
I = np.zeros(0, np.int64)
# (a) no constraint rows
case('m=0', dict(n=3, m=0, j_row=I, j_col=I, dE=e, df=np.array([1.0, -2.0, 0.0]), f=0.0, E=e, x_k=np.zeros(3), c_lb=e, c_ub=e, v_lb=-np.ones(3), v_ub=np.ones(3), delta=0.4))
# (b) README one-variable problem at x=0: min x^2+x s.t. x^2-x = 2
case('1-var', dict(n=1, m=1, j_row=np.array([1]), j_col=np.array([1]), dE=np.array([-1.0]), df=np.array([1.0]), f=0.0, E=np.array([0.0]), x_k=np.zeros(1), c_lb=np.array([2.0]), c_ub=np.array([2.0]), v_lb=np.array([-INF]), v_ub=np.array([INF]), delta=1000.0))
# (c) rows without any Jacobian entry
case('nnz=0', dict(n=2, m=2, j_row=I, j_col=I, dE=e, df=np.array([1.0, 1.0]), f=0.0, E=np.array([0.5, -0.5]), x_k=np.zeros(2), c_lb=np.array([0.0, -INF]), c_ub=np.array([INF, 0.0]), v_lb=-np.ones(2), v_ub=np.ones(2), delta=0.4))
# (d) all variables fixed by their bounds
case('fixed', dict(n=2, m=1, j_row=np.array([1, 1]), j_col=np.array([1, 2]), dE=np.array([1.0, 1.0]), df=np.array([1.0, -1.0]), f=0.0, E=np.array([0.0]), x_k=np.array([0.3, 0.7]), c_lb=np.array([-1.0]), c_ub=np.array([1.0]), v_lb=np.array([0.3, 0.7]), v_ub=np.array([0.3, 0.7]), delta=0.4))
# (e) zero trust region
case('delta=0', dict(n=2, m=1, j_row=np.array([1, 1]), j_col=np.array([1, 2]), dE=np.array([1.0, 1.0]), df=np.array([1.0, -1.0]), f=0.0, E=np.array([0.0]), x_k=np.array([0.3, 0.7]), c_lb=np.array([-1.0]), c_ub=np.array([1.0]), v_lb=-np.ones(2), v_ub=np.ones(2), delta=0.0))
# (f) all entries duplicates of one position, cancelling to an exact zero
case('cancel', dict(n=2, m=1, j_row=np.array([1, 1, 1]), j_col=np.array([1, 1, 2]), dE=np.array([1.0, -1.0, 2.0]), df=np.array([1.0, 1.0]), f=0.0, E=np.array([0.1]), x_k=np.zeros(2), c_lb=np.array([0.0]), c_ub=np.array([0.0]), v_lb=-np.ones(2), v_ub=np.ones(2), delta=0.4))
