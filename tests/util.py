"""Shared generators for the parity tests (seeded, deterministic)."""
import numpy as np

INF = np.inf


def random_subproblem(seed, n, m, density=1.0, dup_frac=0.0, n_range=0, infeasible=False, delta=0.4):
    """A random instance of the data `sub_optimize!` sees (subproblem.jl:229-246): COO Jacobian with
    optional duplicate entries, mixed EQ / >= / <= / range rows, finite variable bounds, a point x_k."""
    rng = np.random.default_rng(seed)
    mask = rng.random((m, n)) < density
    mask[np.arange(m), rng.integers(0, n, m)] = True
    rows, cols = np.nonzero(mask)
    vals = rng.standard_normal(len(rows)) / np.sqrt(max(1.0, density * n))
    ndup = int(dup_frac * len(rows))
    if ndup:
        pick = rng.integers(0, len(rows), ndup)
        rows = np.concatenate([rows, rows[pick]]); cols = np.concatenate([cols, cols[pick]])
        vals = np.concatenate([vals, 0.25 * rng.standard_normal(ndup)])
        perm = rng.permutation(len(rows))
        rows, cols, vals = rows[perm], cols[perm], vals[perm]
    J = np.zeros((m, n))
    np.add.at(J, (rows, cols), vals)
    x_k = rng.uniform(-0.5, 0.5, n)
    v_lb = -np.ones(n); v_ub = np.ones(n)
    p_star = rng.uniform(-0.3, 0.3, n) * min(1.0, delta / 0.4)
    E = rng.standard_normal(m) * 0.1                       # current constraint values b
    act = E + J @ p_star                                    # linearised value at a feasible step
    c_lb = np.full(m, -INF); c_ub = np.full(m, INF)
    meq = m // 3
    c_lb[:meq] = act[:meq]; c_ub[:meq] = act[:meq]
    k = (m - meq) // 2
    c_ub[meq:meq + k] = act[meq:meq + k] + rng.uniform(0, 0.05, k)
    c_lb[meq + k:] = act[meq + k:] - rng.uniform(0, 0.05, m - meq - k)
    for i in range(n_range):                                # two-sided rows at the end
        r = m - 1 - i
        c_lb[r] = act[r] - 0.02; c_ub[r] = act[r] + 0.03
    if infeasible:
        c_lb[0] = c_ub[0] = act[0] + 50.0
    df = rng.standard_normal(n)
    return dict(n=n, m=m, j_row=rows + 1, j_col=cols + 1, dE=vals, df=df, f=0.3, E=E, x_k=x_k,
                c_lb=c_lb, c_ub=c_ub, v_lb=v_lb, v_ub=v_ub, delta=delta, J=J)


def equality_rich_subproblem(seed, n=300, neq=260, nineq=160, per_row=4, delta=0.6):
    """A sparse sub-problem with the row structure of the ACOPF configurations: many equality rows (their null space is small),
    fewer inequality rows, a few non-zeros per row - the shape on which the solver eliminates the equality rows
    (null-space form of the interior-point Newton system, active-set solves in reduced coordinates)."""
    rng = np.random.default_rng(seed)
    m = neq + nineq
    rows = np.repeat(np.arange(m), per_row)
    cols = np.concatenate([rng.choice(n, per_row, replace=False) for _ in range(m)])
    # every equality row owns one column (a permuted identity keeps the equality block of full row rank)
    own = np.arange(neq) * per_row
    cols[own] = rng.permutation(n)[:neq]
    vals = rng.standard_normal(len(rows)) * 0.5
    vals[own] = 1.0 + rng.random(neq)
    J = np.zeros((m, n))
    np.add.at(J, (rows, cols), vals)
    x_k = rng.uniform(-0.3, 0.3, n)
    v_lb = -np.ones(n); v_ub = np.ones(n)
    p_star = rng.uniform(-0.2, 0.2, n)
    E = rng.standard_normal(m) * 0.05
    act = E + J @ p_star
    c_lb = np.full(m, -INF); c_ub = np.full(m, INF)
    c_lb[:neq] = act[:neq]; c_ub[:neq] = act[:neq]
    k = nineq // 2
    c_ub[neq:neq + k] = act[neq:neq + k] + rng.uniform(0, 0.05, k)
    c_lb[neq + k:] = act[neq + k:] - rng.uniform(0, 0.05, nineq - k)
    df = rng.standard_normal(n)
    return dict(n=n, m=m, j_row=rows + 1, j_col=cols + 1, dE=vals, df=df, f=0.1, E=E, x_k=x_k,
                c_lb=c_lb, c_ub=c_ub, v_lb=v_lb, v_ub=v_ub, delta=delta, J=J, p_star=p_star)


def oracle_solve(sp, feasibility=False, qp=None):
    from oracle.subproblem import QpData, QpModel, compute_jacobian_matrix
    A, stored = compute_jacobian_matrix(sp['m'], sp['n'], sp['j_row'] - 1, sp['j_col'] - 1, sp['dE'])
    data = QpData(sp['df'], sp['f'], A, sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub'], stored)
    if qp is None:
        qp = QpModel(data, sp['j_row'], sp['j_col'])
    else:
        qp.data = data
    out = qp.sub_optimize(sp['x_k'], sp['delta'], feasibility)
    return qp, out


def hip_solve(sp, feasibility=False, opt=None, device=0):
    from activesetmethods_amd.subproblem import QpData, HipSubOptimizer
    data = QpData(sp['df'], sp['f'], sp['dE'], sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub'])
    if opt is None:
        opt = HipSubOptimizer(data, sp['j_row'], sp['j_col'], device=device)
    else:
        opt.data = data
    out = opt.sub_optimize(sp['x_k'], sp['delta'], feasibility)
    return opt, out


def rel_err(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.abs(a - b).max(initial=0.0) / max(1.0, np.abs(b).max(initial=0.0)))


def edge_case_subproblems():
    """Degenerate shapes of the sub-LP boundary: no rows, one variable (the README problem at x = 0), rows without
    Jacobian entries, all variables fixed, zero trust region, duplicates cancelling to a stored exact zero."""
    e = np.zeros(0); I = np.zeros(0, np.int64)
    one = lambda **kw: dict(f=0.0, **kw)
    return {
        "no_rows": one(n=3, m=0, j_row=I, j_col=I, dE=e, df=np.array([1.0, -2.0, 0.0]), E=e, x_k=np.zeros(3), c_lb=e, c_ub=e,
                       v_lb=-np.ones(3), v_ub=np.ones(3), delta=0.4),
        "one_var": one(n=1, m=1, j_row=np.array([1]), j_col=np.array([1]), dE=np.array([-1.0]), df=np.array([1.0]), E=np.array([0.0]),
                       x_k=np.zeros(1), c_lb=np.array([2.0]), c_ub=np.array([2.0]), v_lb=np.array([-INF]), v_ub=np.array([INF]), delta=1000.0),
        "no_entries": one(n=2, m=2, j_row=I, j_col=I, dE=e, df=np.array([1.0, 1.0]), E=np.array([0.5, -0.5]), x_k=np.zeros(2),
                          c_lb=np.array([0.0, -INF]), c_ub=np.array([INF, 0.0]), v_lb=-np.ones(2), v_ub=np.ones(2), delta=0.4),
        "all_fixed": one(n=2, m=1, j_row=np.array([1, 1]), j_col=np.array([1, 2]), dE=np.array([1.0, 1.0]), df=np.array([1.0, -1.0]),
                         E=np.array([0.0]), x_k=np.array([0.3, 0.7]), c_lb=np.array([-1.0]), c_ub=np.array([1.0]),
                         v_lb=np.array([0.3, 0.7]), v_ub=np.array([0.3, 0.7]), delta=0.4),
        "zero_radius": one(n=2, m=1, j_row=np.array([1, 1]), j_col=np.array([1, 2]), dE=np.array([1.0, 1.0]), df=np.array([1.0, -1.0]),
                           E=np.array([0.0]), x_k=np.array([0.3, 0.7]), c_lb=np.array([-1.0]), c_ub=np.array([1.0]),
                           v_lb=-np.ones(2), v_ub=np.ones(2), delta=0.0),
        "cancelling_duplicates": one(n=2, m=1, j_row=np.array([1, 1, 1]), j_col=np.array([1, 1, 2]), dE=np.array([1.0, -1.0, 2.0]),
                                     df=np.array([1.0, 1.0]), E=np.array([0.1]), x_k=np.zeros(2), c_lb=np.array([0.0]), c_ub=np.array([0.0]),
                                     v_lb=-np.ones(2), v_ub=np.ones(2), delta=0.4),
    }


EDGE_CASE_ANSWERS = {      # worked by hand from the LP each case poses
    "no_rows": ([-0.4, 0.4, 0.0], []),
    "one_var": ([-2.0], [-1.0]),                 # -p = 2 ; df - J'lambda = 1 - (-1)(-1) = 0
    "no_entries": ([-0.4, -0.4], [0.0, 0.0]),
    "all_fixed": ([0.0, 0.0], [0.0]),
    "zero_radius": ([0.0, 0.0], [0.0]),
    "cancelling_duplicates": ([-0.4, -0.05], [0.5]),   # 2 p2 = -0.1 ; 1 - 2 lambda = 0
}


def banded_subproblem(seed, n=400, m=600, neq=120, nrange=30, width=6, per_row=4, delta=0.5, infeasible=False):
    """A sparse sub-problem whose rows couple locally (row i touches columns near i n / m), stored in a random row order: the coupling
    graph has a small bandwidth that only a reordering finds - the shape on which the library factors banded matrices (reverse
    Cuthill-McKee order of the rows / columns).  m >= 256 rows so that the order is used."""
    rng = np.random.default_rng(seed)
    shuffle = rng.permutation(m)                            # row i of the hidden chain is stored as row shuffle[i]
    rows, cols, vals = [], [], []
    for i in range(m):
        c0 = int(i * n / m)
        cand = np.arange(max(0, c0 - width), min(n, c0 + width + 1))
        cs = rng.choice(cand, min(per_row, len(cand)), replace=False)
        for c in cs:
            rows.append(shuffle[i]); cols.append(int(c)); vals.append(rng.standard_normal() * 0.5 + (1.5 if c == c0 else 0.0))
    rows, cols, vals = np.array(rows), np.array(cols), np.array(vals)
    order = np.lexsort((cols, rows))
    rows, cols, vals = rows[order], cols[order], vals[order]
    J = np.zeros((m, n))
    np.add.at(J, (rows, cols), vals)
    x_k = rng.uniform(-0.3, 0.3, n)
    v_lb = -np.ones(n); v_ub = np.ones(n)
    p_star = rng.uniform(-0.2, 0.2, n)
    E = rng.standard_normal(m) * 0.05
    act = E + J @ p_star
    kinds = rng.permutation(m)
    c_lb = np.full(m, -INF); c_ub = np.full(m, INF)
    eq = kinds[:neq]; rg = kinds[neq:neq + nrange]; rest = kinds[neq + nrange:]
    c_lb[eq] = act[eq]; c_ub[eq] = act[eq]
    c_lb[rg] = act[rg] - rng.uniform(0, 0.1, len(rg)); c_ub[rg] = act[rg] + rng.uniform(0, 0.1, len(rg))
    half = len(rest) // 2
    c_ub[rest[:half]] = act[rest[:half]] + rng.uniform(0, 0.05, half)
    c_lb[rest[half:]] = act[rest[half:]] - rng.uniform(0, 0.05, len(rest) - half)
    if infeasible:                                          # contradictory bounds on a few rows: the normal-phase LP is infeasible
        bad = rest[:8]
        c_ub[bad] = act[bad] - 5.0
    df = rng.standard_normal(n)
    return dict(n=n, m=m, j_row=rows + 1, j_col=cols + 1, dE=vals, df=df, f=0.1, E=E, x_k=x_k,
                c_lb=c_lb, c_ub=c_ub, v_lb=v_lb, v_ub=v_ub, delta=delta, J=J, p_star=p_star)


def random_function_model(seed, n=30, sense="MIN_SENSE"):
    """A random model in the MOI wrapper's six lists (affine and quadratic functions, duplicates, diagonal quadratic terms)."""
    from activesetmethods_amd.moi_evaluator import FunctionModel, ScalarFunction
    rng = np.random.default_rng(seed)
    fm = FunctionModel(n, -np.ones(n), np.ones(n))
    fm.sense = sense

    def func(quad):
        aff = [(float(rng.standard_normal()), int(rng.integers(1, n + 1))) for _ in range(int(rng.integers(0, 6)))]
        q = []
        if quad:
            for _ in range(int(rng.integers(1, 5))):
                a, b = int(rng.integers(1, n + 1)), int(rng.integers(1, n + 1))
                if rng.random() < 0.4:
                    b = a
                q.append((float(rng.standard_normal()), a, b))
        return ScalarFunction(float(rng.standard_normal()), aff, q)
    for kind in ("le", "ge", "eq"):
        for _ in range(int(rng.integers(1, 5))):
            fm.add_constraint(func(False), kind, float(rng.standard_normal()))
        for _ in range(int(rng.integers(1, 5))):
            fm.add_constraint(func(True), kind, float(rng.standard_normal()))
    fm.objective = func(True)
    return fm


def oracle_wrapper_model(fm):
    """The plain-dict model of oracle/moi_eval.py from the RAW data of a FunctionModel (constants and term lists as they were given to it -
    nothing of the product's evaluation or flattening code is used): the independent checker of rows a2 / f3."""
    def fn(f):
        return {"constant": float(f.constant), "affine": [(float(c), int(v)) for c, v in f.affine],
                "quadratic": [(float(c), int(a), int(b)) for c, a, b in f.quadratic]}
    m = {"n": fm.n, "sense": fm.sense, "objective": None if fm.objective is None else fn(fm.objective)}
    for name in ("linear_le", "linear_ge", "linear_eq", "quadratic_le", "quadratic_ge", "quadratic_eq"):
        m[name] = [fn(f) for f, _ in getattr(fm, name)]
    m["nlp"] = None
    if fm.nlp is not None:
        blk = fm.nlp
        m["nlp"] = {"m": blk.m, "pattern": list(zip(blk.rows.tolist(), blk.cols.tolist())),
                    "eval_g": lambda x: blk.eval_g(np.asarray(x, float), np.zeros(blk.m)),
                    "eval_jac": lambda x: blk.eval_jac_g(np.asarray(x, float), np.zeros(len(blk.rows)))}
    return m


def oracle_evaluate(om, x):
    """(f, grad f, g, Jacobian values, j_str) of a wrapper model through oracle/moi_eval.py."""
    from oracle import moi_eval as W
    xs = [float(v) for v in x]
    j_str = W.jacobian_structure(om)
    m = W.nlp_constraint_offset(om) + (om["nlp"]["m"] if om["nlp"] is not None else 0)
    g = W.eval_constraint(om, [0.0] * m, xs)
    vals = W.eval_constraint_jacobian(om, [0.0] * len(j_str), xs)
    grad = W.eval_objective_gradient(om, [0.0] * om["n"], xs)
    return W.eval_objective(om, xs), np.array(grad), np.array(g, float), np.array(vals, float), j_str
