"""Oracle restatement of the sub-problem layer (TEST INFRASTRUCTURE - see oracle/__init__.py).

Follows, function by function:
  compute_jacobian_matrix            src/algorithms/common.jl:12-20
  QpData / LpData                    src/algorithms/subproblem.jl:3-14, src/algorithms/slp.jl:8-21
  QpModel + create_model!            src/algorithms/subproblem.jl:16-215
  sub_optimize!(qp, x_k, Δ, feas)    src/algorithms/subproblem.jl:229-542
The MOI model object the reference mutates call by call is represented by the dense LP of
oracle/lp_solver.py; the LP solve (subproblem.jl:490) is `lp_solver.solve_lp`.
All indices in this file are 0-based; `j_row/j_col` arrive 1-based as in the reference (Int64 tuples,
src/MOI_wrapper.jl:726-746) and are shifted once in `QpModel.__init__`.
"""
import numpy as np
from . import lp_solver as L

INF = np.inf


def compute_jacobian_matrix(m, n, j_row, j_col, dE):
    """common.jl:12-20 - `J = spzeros(m,n); J[r,c] += dE[k]` in j_str order (duplicates accumulate in
    that order).  Returns (J dense m x n, stored m x n bool): `stored[r,c]` reproduces which entries a
    Julia SparseMatrixCSC holds after the loop - an entry is created the first time a non-zero value is
    written to an unstored slot (setindex! drops zeros written to unstored slots), and stays stored
    afterwards even if it later becomes 0.0.  Only subproblem.jl:448-457 (range rows) observes it."""
    J = np.zeros((m, n))
    stored = np.zeros((m, n), bool)
    for k in range(len(dE)):
        r, c = j_row[k], j_col[k]
        J[r, c] = J[r, c] + dE[k]          # unstored slots read as 0.0, exactly as getindex does
        if dE[k] != 0.0 or stored[r, c]:
            stored[r, c] = True
        # (an unstored slot receiving 0.0 stays unstored and J[r,c] stays 0.0)
    return J, stored


class QpData:
    """subproblem.jl:3-14 (Q is always `nothing` on this path, slp.jl:12)."""

    def __init__(self, c, c0, A, b, c_lb, c_ub, v_lb, v_ub, stored=None):
        self.c, self.c0, self.A, self.b = c, c0, A, b
        self.c_lb, self.c_ub, self.v_lb, self.v_ub = c_lb, c_ub, v_lb, v_ub
        self.stored = stored if stored is not None else (A != 0.0)


def row_kind(c_lb, c_ub):
    """Row classification of create_model! (subproblem.jl:143-197): 0 EQ, 2 range (GE + adj LE row),
    +1 lower only (GE), -1 upper only (LE), 9 free (the reference adds no row: unsupported)."""
    if c_lb == c_ub:
        return 0
    if c_lb != -INF and c_ub != INF and c_lb < c_ub:
        return 2
    if c_lb != -INF:
        return 1
    if c_ub != INF:
        return -1
    return 9


class QpModel:
    """subproblem.jl:16-49 + create_model! (51-215): the persistent LP skeleton.

    Row layout: rows 0..m-1 are the reference's `constr[1..m]`; row m+k is the extra `<=` row of the
    k-th range constraint `adj[k]` (subproblem.jl:200-214).  Slack layout (subproblem.jl:83-112): one
    slack per row, two if both bounds are finite (equalities included)."""

    def __init__(self, data, j_row, j_col):
        self.data = data
        self.j_row = np.asarray(j_row, np.int64) - 1
        self.j_col = np.asarray(j_col, np.int64) - 1
        m, n = len(data.c_lb), len(data.c)
        assert n > 0 and m >= 0
        self.m, self.n = m, n
        rc = [[] for _ in range(m)]
        for r_, c_ in zip(self.j_row, self.j_col):
            rc[r_].append(int(c_))
        self.row_cols = [np.unique(np.array(v, np.int64)) for v in rc]
        self.kind = np.array([row_kind(data.c_lb[i], data.c_ub[i]) for i in range(m)], np.int64)
        if (self.kind == 9).any():
            raise ValueError("free constraint rows are not representable (subproblem.jl:143-197 adds no row)")
        self.adj = np.nonzero(self.kind == 2)[0]
        self.M = m + len(self.adj)
        # row order of the factorisations (lp_solver.row_order): all M rows in reverse Cuthill-McKee order of their coupling graph when the
        # pattern is sparse and that order has a bandwidth below M / 2 - the HIP library's factorisations stop at the band; this dense
        # restatement only shares the order (the pivot guard drops the LATER of two dependent rows)
        self.row_pos = L.row_order(self.row_cols + [self.row_cols[v] for v in self.adj], n)
        # LP row types
        rtype = np.zeros(self.M, np.int64)
        rtype[:m] = np.where(self.kind == 0, 0, np.where(self.kind == -1, -1, 1))
        rtype[m:] = -1
        self.rtype = rtype
        # slacks: (row, coeff) ; nslack[i] in {1,2}
        self.nslack = np.where((data.c_lb > -INF) & (data.c_ub < INF), 2, 1)
        srow, scoef, sown = [], [], []
        adjpos = {int(v): k for k, v in enumerate(self.adj)}
        for i in range(m):
            k = self.kind[i]
            if k == 0:
                srow += [i, i]; scoef += [1.0, -1.0]; sown += [(i, 0), (i, 1)]
            elif k == 2:
                srow += [i, m + adjpos[i]]; scoef += [1.0, -1.0]; sown += [(i, 0), (i, 1)]
            elif k == 1:
                srow += [i]; scoef += [1.0]; sown += [(i, 0)]
            else:
                srow += [i]; scoef += [-1.0]; sown += [(i, 0)]
        self.srow = np.array(srow, np.int64)
        self.scoef = np.array(scoef, float)
        self.sown = sown
        # coefficients of the adj rows persist across calls (subproblem.jl:448-457 refreshes only the
        # entries currently stored in A[val,:]); create_model! starts them with slack terms only.
        self.A_adj = np.zeros((len(self.adj), n))
        self.warm = {False: None, True: None}     # last accepted active set per phase
        # adaptive solver decisions per phase (lp_solver.solve_scaled); restoration LPs (min sum of slacks) usually have a
        # non-unique optimum, so that phase starts with the polish from the interior-point iterate preferred
        self.hint = {False: {}, True: {'prefer_ref': True}}

    # ------------------------------------------------------------------ sub_optimize!
    def build_lp(self, x_k, Delta, feasibility):
        """Everything of subproblem.jl:248-484 - returns the LP that `MOI.optimize!` would see."""
        d, m, n = self.data, self.m, self.n
        b = d.b.copy()                                               # :248
        # variable bounds (:427-434)
        ub = np.minimum(Delta, d.v_ub - x_k)
        lb = np.maximum(-Delta, d.v_lb - x_k)
        # row coefficients (:438-457)
        A = np.zeros((self.M, n))
        A[:m] = d.A
        for k, val in enumerate(self.adj):
            st = d.stored[val]
            self.A_adj[k, st] = d.A[val, st]
        A[m:] = self.A_adj
        if feasibility:
            viol = np.where(d.b > d.c_ub, d.c_ub - d.b, np.where(d.b < d.c_lb, d.c_lb - d.b, 0.0))   # :289-294
            b = b - np.abs(viol)                                                                    # :295
            slo = []
            for i in range(m):                                                                        # :298-381
                if self.nslack[i] == 2:
                    slo += [0.0, viol[i]] if viol[i] < 0 else [-viol[i], 0.0]
                else:
                    slo += [-abs(viol[i])]
            slo = np.array(slo, float)
            q = np.zeros(n)                                                                           # :252-263
            w = np.ones(len(self.srow))                                                               # :266-272
            srow, scoef = self.srow, self.scoef
        else:
            q = d.c.copy()                                                                            # :385-396
            srow = scoef = w = slo = None           # slacks fixed to 0 (EqualTo(0.0), :418-423)
        # right-hand sides (:461-484)
        r = np.empty(self.M)
        c_ub = d.c_ub - b
        c_lb = d.c_lb - b
        r[:m] = np.where(self.kind == -1, c_ub, c_lb)
        r[m:] = c_ub[self.adj]
        lp = L.LP(q, A, self.rtype, r, lb, ub, srow, scoef, w, slo)
        lp.row_cols = self.row_cols                 # structural pattern (j_row, j_col) of the m constraint rows (the adj rows are inequalities)
        lp.row_pos = self.row_pos
        return lp

    def sub_optimize(self, x_k, Delta, feasibility=False):
        """subproblem.jl:229-542.  Returns (Xsol, lambda, mult_x_U, mult_x_L, p_slack, status, info)."""
        d, m, n = self.data, self.m, self.n
        assert len(d.c) == n and len(d.c_lb) == m and len(d.c_ub) == m
        assert len(d.v_lb) == n and len(d.v_ub) == n and len(x_k) == n
        lp = self.build_lp(x_k, Delta, feasibility)
        out = L.solve_lp(lp, self.warm[bool(feasibility)], self.hint[bool(feasibility)])
        status = out['status']
        Xsol = np.zeros(n); lam = np.zeros(m); mult_x_U = np.zeros(n); mult_x_L = np.zeros(n)
        p_slack = {}
        if status == L.OPTIMAL:
            prev = self.warm[bool(feasibility)]
            self.hint[bool(feasibility)]['stable'] = bool(prev is not None and all(np.array_equal(a, b) for a, b in zip(prev, out['sets'])))
            self.warm[bool(feasibility)] = out['sets']
            rowst, bst, sst = out['sets']
            Xsol[:] = out['p']                                                      # :502
            s = out['s'] if feasibility else np.zeros(len(self.srow))
            for k, (i, pos) in enumerate(self.sown):                                # :503-505
                p_slack.setdefault(i, []).append(float(s[k]))
            y = out['y'].copy()
            # multipliers with the sign their row type admits (GLPK returns sign-feasible duals)
            y = np.where(self.rtype == 1, np.maximum(y, 0.0), np.where(self.rtype == -1, np.minimum(y, 0.0), y))
            lam[:] = y[:m]                                                          # :510-512
            for k, val in enumerate(self.adj):                                      # :513-515
                lam[val] += y[m + k]
            z = out['z']
            fixed = lp.ub <= lp.lb
            mult_x_L[:] = np.where(bst < 0, np.maximum(z, 0.0), 0.0)                 # :519-520
            mult_x_U[:] = np.where(bst > 0, np.minimum(z, 0.0), 0.0)
            mult_x_U[:] = np.where(fixed, np.minimum(z, 0.0), mult_x_U)            # a fixed column reports both halves of its reduced cost
            mult_x_L[:] = np.where(fixed, np.maximum(z, 0.0), mult_x_L)
            mult_x_U[Xsol < d.v_ub - x_k] = 0.0                                      # :522-529
            mult_x_L[Xsol > d.v_lb - x_k] = 0.0
        # INFEASIBLE -> all zero (:532-536); other statuses: outputs undefined in the reference (:537-538)
        return Xsol, lam, mult_x_U, mult_x_L, p_slack, status, out
