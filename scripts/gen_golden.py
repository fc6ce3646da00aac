#!/usr/bin/env python
"""Generates tests/golden/lp_highs.npz: seeded, non-degenerate LPs in the sub-LP form of
src/algorithms/subproblem.jl together with their optima from SciPy's HiGHS dual simplex
(scipy.optimize.linprog, method="highs-ds").  HiGHS is an independent LP code used as a cross-check;
it is NOT the reference's solver (GLPK, not available offline) - see SURVEY.md section 8c.

Run in the build container:  python scripts/gen_golden.py
"""
import os
import sys

import numpy as np
from scipy.optimize import linprog

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INF = np.inf


def make_lp(seed, n, m, meq):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, n)) / np.sqrt(n)
    z = rng.uniform(-0.3, 0.3, n)
    Az = A @ z
    rtype = np.zeros(m, np.int64)
    r = Az.copy()
    k = (m - meq) // 2
    rtype[meq:meq + k] = -1
    r[meq:meq + k] += rng.uniform(0, 0.1, k)
    rtype[meq + k:] = 1
    r[meq + k:] -= rng.uniform(0, 0.1, m - meq - k)
    lb = np.full(n, -0.4)
    ub = np.full(n, 0.4)
    q = rng.standard_normal(n)
    return q, A, rtype, r, lb, ub


def highs(q, A, rtype, r, lb, ub):
    eq = rtype == 0
    le = rtype == -1
    ge = rtype == 1
    A_ub = np.vstack([A[le], -A[ge]])
    b_ub = np.concatenate([r[le], -r[ge]])
    res = linprog(q, A_ub=A_ub, b_ub=b_ub, A_eq=A[eq], b_eq=r[eq], bounds=list(zip(lb, ub)), method="highs-ds")
    assert res.status == 0
    y = np.zeros(len(r))
    y[eq] = res.eqlin.marginals
    y[le] = res.ineqlin.marginals[:le.sum()]
    y[ge] = -res.ineqlin.marginals[le.sum():]
    z = res.lower.marginals + res.upper.marginals
    return res.x, y, z, res.fun


def main():
    out = {}
    for idx, (seed, n, m, meq) in enumerate([(101, 12, 8, 3), (102, 40, 25, 10), (103, 90, 60, 20), (104, 60, 80, 25)]):
        q, A, rtype, r, lb, ub = make_lp(seed, n, m, meq)
        x, y, z, fun = highs(q, A, rtype, r, lb, ub)
        for k, v in dict(q=q, A=A, rtype=rtype, r=r, lb=lb, ub=ub, x=x, y=y, z=z, obj=np.array(fun)).items():
            out["lp%d_%s" % (idx, k)] = v
    out["count"] = np.array(4)
    path = os.path.join(ROOT, "tests", "golden", "lp_highs.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
