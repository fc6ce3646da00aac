"""Sparse statement of the sub-LP of `sub_optimize!` (src/algorithms/subproblem.jl:229-484) for third-party LP codes
(TEST INFRASTRUCTURE - see oracle/__init__.py; used by bench.py's `cpu_baseline` leg and by tests/).

Same formulation as `oracle.subproblem.QpModel.build_lp` (rows, slack layout, the literal `b -= abs(viol)` shift and
the slack lower bounds of the restoration phase), but the Jacobian stays a scipy.sparse matrix, so the ACOPF LPs
(fill 0.03 %) can be handed to a sparse simplex code - the natural CPU method for them, and the kind of code the
reference drives (GLPK).  The stale-coefficient rule of range rows (subproblem.jl:448-457) is not modelled: it only
matters when a stored Jacobian entry becomes exactly 0.0, which the benchmark problems never produce.
"""
import numpy as np
import scipy.sparse as sp

INF = np.inf


def build(n, m, j_row, j_col, dE, df, E, c_lb, c_ub, v_lb, v_ub, x_k, Delta, feasibility):
    """Returns dict(c, A_ub, b_ub, A_eq, b_eq, bounds, n, ns) for scipy.optimize.linprog: variables = [p (n), slacks]."""
    J = sp.coo_matrix((np.asarray(dE, float), (np.asarray(j_row) - 1, np.asarray(j_col) - 1)), shape=(m, n)).tocsr()
    c_lb = np.asarray(c_lb, float); c_ub = np.asarray(c_ub, float); E = np.asarray(E, float)
    eq = c_lb == c_ub
    rng = (c_lb > -INF) & (c_ub < INF) & (c_lb < c_ub)
    lo = (c_lb > -INF) & ~eq & ~rng
    up = (c_ub < INF) & ~eq & ~rng
    b = E.copy()
    lb = np.maximum(-Delta, np.asarray(v_lb, float) - x_k)                  # subproblem.jl:427-434
    ub = np.minimum(Delta, np.asarray(v_ub, float) - x_k)
    # slack columns (subproblem.jl:83-112): one per row, two when both bounds are finite
    nsl = np.where((c_lb > -INF) & (c_ub < INF), 2, 1)
    first = np.concatenate([[0], np.cumsum(nsl)[:-1]]) if m else np.zeros(0, np.int64)
    ns = int(nsl.sum()) if feasibility else 0
    slo = np.zeros(ns)
    if feasibility:
        viol = np.where(E > c_ub, c_ub - E, np.where(E < c_lb, c_lb - E, 0.0))   # :289-294
        b = b - np.abs(viol)                                                     # :295
        two = nsl == 2
        slo[first[two]] = np.where(viol[two] < 0, 0.0, -viol[two])               # :298-381
        slo[first[two] + 1] = np.where(viol[two] < 0, viol[two], 0.0)
        slo[first[~two]] = -np.abs(viol[~two])
    rhs_lb, rhs_ub = c_lb - b, c_ub - b

    def slack_block(rows, col_of_row, coef):
        if not feasibility:
            return None
        k = len(rows)
        return sp.coo_matrix((np.full(k, coef), (np.arange(k), col_of_row)), shape=(k, ns)).tocsr()

    def with_slacks(Jr, blocks):
        if not feasibility:
            return Jr
        S = blocks[0]
        for B in blocks[1:]:
            S = S + B
        return sp.hstack([Jr, S]).tocsr()

    r_eq = np.nonzero(eq)[0]
    A_eq = with_slacks(J[r_eq], [slack_block(r_eq, first[r_eq], 1.0), slack_block(r_eq, first[r_eq] + 1, -1.0)])        # s1 - s2 + A p = c - b
    b_eq = rhs_lb[r_eq]
    r_ge = np.nonzero(lo | rng)[0]                                                                                        # s1 + A p >= c_lb - b
    A_ge = with_slacks(J[r_ge], [slack_block(r_ge, first[r_ge], 1.0)])
    r_le = np.nonzero(up)[0]                                                                                              # -s1 + A p <= c_ub - b
    A_le = with_slacks(J[r_le], [slack_block(r_le, first[r_le], -1.0)])
    r_adj = np.nonzero(rng)[0]                                                                                            # -s2 + A p <= c_ub - b
    A_adj = with_slacks(J[r_adj], [slack_block(r_adj, first[r_adj] + 1, -1.0)])
    A_ub = sp.vstack([-A_ge, A_le, A_adj]).tocsr()
    b_ub = np.concatenate([-rhs_lb[r_ge], rhs_ub[r_le], rhs_ub[r_adj]])
    cost = np.concatenate([np.zeros(n), np.ones(ns)]) if feasibility else np.asarray(df, float).copy()
    bounds = np.c_[np.concatenate([lb, slo]), np.concatenate([ub, np.full(ns, INF)])]
    return dict(c=cost, A_ub=A_ub, b_ub=b_ub, A_eq=A_eq, b_eq=b_eq, bounds=bounds, n=n, ns=ns)


def solve_highs(lp, method="highs-ds"):
    """scipy.optimize.linprog on the LP of `build`.  Returns (status, objective, p, seconds, simplex iterations); status in
    the MOI codes of include/asm_hip.h (1 OPTIMAL, 2 INFEASIBLE, 4 OTHER)."""
    import time
    from scipy.optimize import linprog
    t0 = time.perf_counter()
    res = linprog(lp['c'], A_ub=lp['A_ub'], b_ub=lp['b_ub'], A_eq=lp['A_eq'], b_eq=lp['b_eq'], bounds=lp['bounds'], method=method)
    dt = time.perf_counter() - t0
    st = 1 if res.status == 0 else (2 if res.status == 2 else 4)
    return st, (float(res.fun) if res.status == 0 else None), (res.x[:lp['n']] if res.status == 0 else None), dt, int(getattr(res, 'nit', 0))
