"""CPU suite: pins the oracle (the checker of every GPU parity test) against
  - the known answers the reference's own tests hold for this path (test/runtests.jl:11-13, the two LP
    snapshots test/sublp_org.lp / test/sublp.lp whose data is kept in tests/golden/sublp_snapshots.json),
  - SciPy/HiGHS optima of seeded LPs (tests/golden/lp_highs.npz, made by scripts/gen_golden.py),
and checks the literal quirks of the formulation."""
import json
import os

import numpy as np
import pytest

from oracle import lp_solver as L
from oracle import slp as O
from oracle.subproblem import QpData, QpModel, compute_jacobian_matrix

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INF = np.inf


def test_lp_snapshots_known_answers():
    d = json.load(open(os.path.join(GOLD, "sublp_snapshots.json")))
    for name in ("sublp_org", "sublp"):
        s = d[name]
        lp = L.LP(s["q"], np.array(s["A"]), s["rtype"], s["r"], s["lb"], s["ub"], s["srow"], s["scoef"], s["w"], np.zeros(8))
        out = L.solve_lp(lp)
        assert out["status"] == L.OPTIMAL
        obj = np.dot(s["q"], out["p"]) + np.dot(s["w"], out["s"])
        assert abs(obj - s["expected_obj"]) < 1e-9
        if "expected_x1" in s:
            assert abs(out["p"][0] - s["expected_x1"]) < 1e-12
        if "expected_dual_c1" in s:
            assert abs(out["y"][2] - s["expected_dual_c1"]) < 1e-9


def test_lp_solver_matches_highs_goldens():
    g = np.load(os.path.join(GOLD, "lp_highs.npz"))
    for i in range(int(g["count"])):
        f = lambda k: g["lp%d_%s" % (i, k)]
        out = L.solve_lp(L.LP(f("q"), f("A"), f("rtype"), f("r"), f("lb"), f("ub")))
        assert out["status"] == L.OPTIMAL and out["stats"]["polished"] == 1
        assert np.abs(out["p"] - f("x")).max() < 1e-9
        assert np.abs(out["y"] - f("y")).max() < 1e-8
        assert np.abs(out["z"] - f("z")).max() < 1e-8
        assert abs(f("q") @ out["p"] - float(f("obj"))) < 1e-9
        rowst, bst, sst = out["sets"]
        assert np.array_equal(bst == -1, np.isclose(f("x"), f("lb"), atol=1e-9))     # identical active bounds
        assert np.array_equal(bst == 1, np.isclose(f("x"), f("ub"), atol=1e-9))


def test_infeasible_lp_detected_with_farkas_certificate():
    g = np.load(os.path.join(GOLD, "lp_highs.npz"))
    f = lambda k: g["lp1_%s" % k]
    r = f("r").copy()
    r[0] += 50.0
    out = L.solve_lp(L.LP(f("q"), f("A"), f("rtype"), r, f("lb"), f("ub")))
    assert out["status"] == L.INFEASIBLE


def test_scaling_is_exact_powers_of_two():
    x = np.array([3.0, 0.7, 0.71, 1e-5, 0.0, -2.0, 1000.0])
    p = L.pow2_round(x)
    assert np.array_equal(p, [4.0, 0.5, 1.0, 2.0 ** -17, 1.0, 1.0, 1024.0])


def test_jacobian_assembly_duplicates_and_storage():
    """common.jl:12-20 `+=` in j_str order; stored-entry semantics observed by subproblem.jl:448-457."""
    rows = np.array([0, 0, 1, 1, 1]); cols = np.array([1, 1, 0, 2, 2])
    J, st = compute_jacobian_matrix(2, 3, rows, cols, np.array([1.0, 2.0, 0.0, 5.0, -5.0]))
    assert J[0, 1] == 3.0 and J[1, 2] == 0.0 and J[1, 0] == 0.0
    assert st[0, 1] and st[1, 2] and not st[1, 0] and not st[0, 0]     # (1,2) stored although it cancelled to 0


def _qp(c_lb, c_ub, E, J, n=2):
    m = len(c_lb)
    rows, cols = np.nonzero(np.ones((m, n)))
    data = QpData(np.ones(n), 0.0, np.asarray(J, float), np.asarray(E, float), np.asarray(c_lb, float), np.asarray(c_ub, float),
                  -10 * np.ones(n), 10 * np.ones(n))
    return QpModel(data, rows + 1, cols + 1)


def test_formulation_row_and_slack_layout():
    """create_model! (subproblem.jl:83-214): 2 slacks when both bounds finite (equalities too), extra <= row per range row."""
    qp = _qp([1.0, -INF, 0.0, 0.0], [1.0, 2.0, INF, 3.0], [0.0, 0.0, 0.0, 0.0], np.ones((4, 2)))
    assert list(qp.kind) == [0, -1, 1, 2] and list(qp.adj) == [3] and qp.M == 5
    assert list(qp.rtype) == [0, -1, 1, 1, -1]
    assert list(qp.nslack) == [2, 1, 1, 2]
    assert list(qp.srow) == [0, 0, 1, 2, 3, 4] and list(qp.scoef) == [1, -1, -1, 1, 1, -1]


def test_restoration_shift_is_abs_viol_in_both_directions():
    """subproblem.jl:289-295: b[i] -= abs(viol) whether the row is violated above or below; slack lower
    bounds (0, viol) / (-viol, 0) for 2-slack rows and -abs(viol) for 1-slack rows (:298-381)."""
    qp = _qp([1.0, -INF, 0.0], [1.0, 2.0, INF], [3.0, 5.0, -4.0], np.ones((3, 2)))
    lp = qp.build_lp(np.zeros(2), 1000.0, True)
    viol = np.array([1.0 - 3.0, 2.0 - 5.0, 0.0 - (-4.0)])           # c_ub-b (above), c_ub-b (above), c_lb-b (below)
    b_shift = np.array([3.0, 5.0, -4.0]) - np.abs(viol)
    assert np.allclose(lp.r, [1.0 - b_shift[0], 2.0 - b_shift[1], 0.0 - b_shift[2]])
    assert np.allclose(lp.slo, [0.0, viol[0], -abs(viol[1]), -abs(viol[2])])
    assert np.all(lp.q == 0) and np.all(lp.w == 1)
    lpn = qp.build_lp(np.zeros(2), 1000.0, False)
    assert lpn.ns == 0 and np.allclose(lpn.r, [1.0 - 3.0, 2.0 - 5.0, 0.0 + 4.0])


def test_trust_region_only_multipliers_are_zeroed():
    """subproblem.jl:522-529: a bound multiplier survives only if the active bound is the variable's own bound."""
    J = np.array([[1.0, 1.0]])
    data = QpData(np.array([1.0, -1.0]), 0.0, J, np.array([0.0]), np.array([-INF]), np.array([10.0]), np.array([-0.05, -5.0]),
                  np.array([5.0, 5.0]))
    qp = QpModel(data, [1, 1], [1, 2])
    X, lam, mU, mL, ps, st, info = qp.sub_optimize(np.zeros(2), 0.1, False)
    assert st == L.OPTIMAL and np.allclose(X, [-0.05, 0.1])
    assert mL[0] == pytest.approx(1.0) and mU[1] == 0.0            # x1 at its own lower bound; x2 only at the trust region


@pytest.mark.parametrize("alg", ["Line Search", "Trust Region"])
def test_toy_known_answer(alg):
    """test/runtests.jl:11-13: X = Y = -1 (rtol 1e-4), LOCALLY_SOLVED; multipliers (0, 1/3, 0, 0) follow
    from df - J'lambda = 0 at the solution (common.jl:38)."""
    from activesetmethods_amd import problems
    pr = problems.toy_problem()
    m = O.Model(pr.n, pr.m, pr.x_L, pr.x_U, pr.g_L, pr.g_U, pr.j_str, pr.eval_f, pr.eval_g, pr.eval_grad_f, pr.eval_jac_g,
                O.Parameters(algorithm=alg))
    m.x[:] = pr.x0
    slp = O.optimize(m)
    assert m.status == 0
    assert np.allclose(m.x, [-1.0, -1.0], rtol=1e-4)
    assert np.allclose(m.mult_g, [0.0, 1.0 / 3.0, 0.0, 0.0], atol=1e-6)
    assert slp.trace[0]["status"] == L.INFEASIBLE and slp.trace[1]["fr"]     # first LP infeasible -> restoration


def test_norm_helpers():
    E = np.array([3.0, -1.0, 0.5]); gL = np.array([0.0, 0.0, 0.5]); gU = np.array([2.0, INF, 0.5])
    x = np.array([2.0, -3.0]); xL = np.array([-1.0, -1.0]); xU = np.array([1.0, 1.0])
    assert O.norm_violations(E, gL, gU, x, xL, xU, 1) == pytest.approx(1 + 1 + 1 + 2)
    assert O.norm_violations(E, gL, gU, x, xL, xU, INF) == pytest.approx(2.0)
    lam = np.array([2.0, -1.0, 7.0])
    c = O.norm_complementarity(E, gL, gU, lam)
    assert c == pytest.approx(2.0 / (1 + np.sqrt(5.0)))              # max|min(3,-1)*2|, |min(-1,inf)*-1| ; eq row skipped


@pytest.mark.parametrize("alg", ["Line Search", "Trust Region"])
def test_acopf_case3_known_objective(alg):
    """test/runtests.jl:18-19 (via test/opf.jl): ACOPF on examples/acopf/case3.m, max_iter 100 ->
    objective 5906.87949 (rtol 1e-3).  Data: tests/golden/case3.json; formulation: activesetmethods_amd/acopf.py."""
    from activesetmethods_amd import acopf
    d = json.load(open(os.path.join(GOLD, "case3.json")))
    pr = acopf.acopf_problem(acopf.case_from_tables(d["baseMVA"], d["bus"], d["gen"], d["gencost"], d["branch"], d["dcline"]), "case3")
    assert (pr.n, pr.m) == (28, 32)
    # analytic Jacobian vs central differences
    x = pr.x0 + 0.05 * np.random.default_rng(0).standard_normal(pr.n)
    J = np.zeros((pr.m, pr.n))
    np.add.at(J, (pr.j_row - 1, pr.j_col - 1), pr.eval_jac_g(x, np.zeros(pr.nnz)))
    for j in range(pr.n):
        e = np.zeros(pr.n); e[j] = 1e-6
        fd = (pr.eval_g(x + e, np.zeros(pr.m)) - pr.eval_g(x - e, np.zeros(pr.m))) / 2e-6
        assert np.abs(fd - J[:, j]).max() < 1e-7
    m = O.Model(pr.n, pr.m, pr.x_L, pr.x_U, pr.g_L, pr.g_U, pr.j_str, pr.eval_f, pr.eval_g, pr.eval_grad_f, pr.eval_jac_g,
                O.Parameters(algorithm=alg, max_iter=100))
    m.x[:] = pr.x0
    O.optimize(m)
    assert m.status == 0
    assert abs(m.obj_val - d["expected_objective"]) <= 1e-3 * d["expected_objective"]


def test_edge_case_shapes_have_their_hand_worked_answers():
    from tests.util import edge_case_subproblems, EDGE_CASE_ANSWERS, oracle_solve
    for name, sp in edge_case_subproblems().items():
        qp, out = oracle_solve(sp)
        assert out[5] == 1, name
        p_ref, lam_ref = EDGE_CASE_ANSWERS[name]
        assert np.allclose(out[0], p_ref, atol=1e-12), name
        assert np.allclose(out[1], lam_ref, atol=1e-12), name


def test_newton_system_forms_agree():
    """The column form (restoration LPs) and the reduced row form (normal phase) are only cheaper ways to precondition
    the same Newton systems: switched off, the LP solver must end on the same active sets and the same point."""
    from oracle import lp_solver as L
    from tests.util import random_subproblem, oracle_solve
    # restoration LP with n well below M -> column form
    sp = random_subproblem(61, 40, 120, 0.3, 0.1, 4, infeasible=True)
    qp, out = oracle_solve(sp)
    assert out[5] == 2
    qp, a = oracle_solve(sp, True, qp)
    assert a[5] == 1 and a[6]['stats']['col_iters'] > 0
    saved = L.COL_MAX_RATIO
    try:
        L.COL_MAX_RATIO = 0.0
        qp2, out2 = oracle_solve(sp)
        qp2, b = oracle_solve(sp, True, qp2)
    finally:
        L.COL_MAX_RATIO = saved
    assert b[5] == 1 and b[6]['stats']['col_iters'] == 0
    assert all(np.array_equal(x, y) for x, y in zip(a[6]['sets'], b[6]['sets']))
    assert np.abs(a[0] - b[0]).max() < 1e-9 and np.abs(a[1] - b[1]).max() < 1e-8
    # sparse normal-phase LP with many slack-dominated inequality rows -> reduced row form (size gate lowered for the test)
    sp = random_subproblem(62, 300, 260, 0.01, 0.0, 0, delta=0.05)
    saved = L.RED_MIN_M
    try:
        L.RED_MIN_M = 64
        qp, a = oracle_solve(sp)
    finally:
        L.RED_MIN_M = saved
    qp, b = oracle_solve(sp)
    assert a[5] == b[5] == 1 and b[6]['stats']['red_iters'] == 0
    assert all(np.array_equal(x, y) for x, y in zip(a[6]['sets'], b[6]['sets']))
    assert np.abs(a[0] - b[0]).max() < 1e-9 and np.abs(a[1] - b[1]).max() < 1e-8
    assert a[6]['stats']['red_iters'] > 0, a[6]['stats']


@pytest.mark.parametrize("seed,infeasible", [(11, False), (12, True), (13, True)])
def test_sparse_lp_statement_agrees_with_oracle_and_highs(seed, infeasible):
    """oracle/sparse_lp.py (the sparse statement handed to HiGHS by bench.py's cpu_baseline) poses the same LP as the
    oracle's dense build_lp: same status, same optimal value in the normal and the restoration phase - HiGHS (an
    independent LP code) and the oracle's own solver agree on it."""
    from oracle import sparse_lp
    from tests.util import random_subproblem, oracle_solve
    sp = random_subproblem(seed, 40, 30, 0.3, 0.2, 3, infeasible=infeasible)
    qp, out = oracle_solve(sp)
    lp = sparse_lp.build(sp['n'], sp['m'], sp['j_row'], sp['j_col'], sp['dE'], sp['df'], sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub'],
                         sp['x_k'], sp['delta'], False)
    st, obj, p, _, _ = sparse_lp.solve_highs(lp)
    assert st == out[5]
    if st == 1:
        assert abs(obj - sp['df'] @ out[0]) <= 1e-9 * max(1.0, abs(obj))
    else:
        qp, out = oracle_solve(sp, True, qp)
        lp = sparse_lp.build(sp['n'], sp['m'], sp['j_row'], sp['j_col'], sp['dE'], sp['df'], sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub'],
                             sp['x_k'], sp['delta'], True)
        st, obj, p, _, _ = sparse_lp.solve_highs(lp)
        assert st == out[5] == 1
        assert abs(obj - sum(sum(v) for v in out[4].values())) <= 1e-8 * max(1.0, abs(obj))


# ----------------------------------------------------------------------------- round 3: null-space form (oracle/lp_solver.py: NullSpace)
def _eq_rich_lp(seed, n=300, neq=260, nineq=160):
    from tests.util import equality_rich_subproblem
    from oracle.subproblem import QpData, QpModel, compute_jacobian_matrix
    sp = equality_rich_subproblem(seed, n, neq, nineq)
    A, stored = compute_jacobian_matrix(sp['m'], sp['n'], sp['j_row'] - 1, sp['j_col'] - 1, sp['dE'])
    qp = QpModel(QpData(sp['df'], sp['f'], A, sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub'], stored), sp['j_row'], sp['j_col'])
    return sp, qp.build_lp(sp['x_k'], sp['delta'], False)


@pytest.mark.parametrize("seed", [81, 84])
def test_null_space_form_agrees_with_row_form_and_highs(seed, monkeypatch):
    """The same LP solved with the equality rows eliminated (k x k Newton systems, active-set solves in reduced coordinates) and with
    the M x M row form: same status and working sets, step and multipliers to 1e-8; optimal value against HiGHS."""
    from oracle import lp_solver as L
    from scipy.optimize import linprog
    sp, lp = _eq_rich_lp(seed)
    assert L.ns_applicable(lp)
    a = L.solve_lp(lp)
    assert a['status'] == L.OPTIMAL and a['stats']['ns_iters'] > 0 and a['stats']['ns_cold'] == 1
    monkeypatch.setattr(L, "ns_applicable", lambda lp_: False)
    b = L.solve_lp(lp)
    assert b['status'] == L.OPTIMAL and b['stats']['ns_iters'] == 0
    assert all(np.array_equal(u, v) for u, v in zip(a['sets'], b['sets']))
    assert np.abs(a['p'] - b['p']).max() <= 1e-8 * max(1.0, np.abs(b['p']).max())
    assert np.abs(a['y'] - b['y']).max() <= 1e-8 * max(1.0, np.abs(b['y']).max())
    e, g, l = lp.rtype == 0, lp.rtype == 1, lp.rtype == -1
    res = linprog(lp.q, A_ub=np.vstack([-lp.A[g], lp.A[l]]), b_ub=np.concatenate([-lp.r[g], lp.r[l]]), A_eq=lp.A[e], b_eq=lp.r[e],
                  bounds=np.c_[lp.lb, lp.ub], method="highs")
    assert res.status == 0 and abs(lp.q @ a['p'] - res.fun) <= 1e-8 * max(1.0, abs(res.fun))


def test_null_space_basis_is_carried_between_lps():
    """The orthonormal basis of one LP, projected onto the next LP's null space, is accepted when the equality rows move a little and
    rejected (fresh selection) when they are replaced; the basis is orthonormal and annihilates the equality rows to 1e-12."""
    from oracle import lp_solver as L
    sp, lp = _eq_rich_lp(85)
    s1, c, rho, kap = L.scale_lp(lp)
    n1 = L.NullSpace(s1)
    assert n1.valid and n1.cold and n1.k == 300 - 260
    assert np.abs(n1.Zt @ n1.Zt.T - np.eye(n1.k)).max() < 1e-12 and np.abs(n1.AEF @ n1.Zt.T).max() < 1e-12
    lp2 = L.LP(lp.q, lp.A * (1.0 + 1e-2 * np.random.default_rng(1).standard_normal(lp.A.shape)), lp.rtype, lp.r, lp.lb, lp.ub)
    s2, _, _, _ = L.scale_lp(lp2)
    n2 = L.NullSpace(s2, n1.J, n1.Zt)
    assert n2.valid and not n2.cold and n2.how == 'basis'
    assert np.abs(n2.Zt @ n2.Zt.T - np.eye(n2.k)).max() < 1e-12 and np.abs(n2.AEF @ n2.Zt.T).max() < 1e-12
    sp3, lp3 = _eq_rich_lp(86)
    s3, _, _, _ = L.scale_lp(lp3)
    n3 = L.NullSpace(s3, n1.J, n1.Zt)                      # unrelated rows: neither the old basis nor the old columns survive
    assert n3.valid and n3.how in ('cold', 'columns')


def test_reduced_active_set_solve_agrees_with_the_full_one():
    """eqp_ns (constraints of the working set on the reduced coordinates) against eqp (Gram matrix of all active rows) on the optimal
    working set: the same point and multipliers."""
    from oracle import lp_solver as L
    sp, lp = _eq_rich_lp(87)
    out = L.solve_lp(lp)
    assert out['status'] == L.OPTIMAL
    s, c, rho, kap = L.scale_lp(lp)
    nsp = L.NullSpace(s)
    zero_p = np.clip(np.zeros(s.n), s.lb, s.ub)
    p1, _, y1, _ = L.eqp(s, out['sets'], zero_p, np.zeros(s.M))
    p2, _, y2, _ = L.eqp_ns(s, nsp, out['sets'])
    assert np.abs(p1 - p2).max() <= 1e-10 * max(1.0, np.abs(p1).max())
    assert np.abs(y1 - y2).max() <= 1e-9 * max(1.0, np.abs(y1).max())


def test_rcm_order_is_a_bandwidth_reducing_permutation():
    """rcm_order on a shuffled banded pattern and on a grid graph: a permutation, the reported bandwidth is the bandwidth of the
    permuted pattern, close to the hidden band (shuffled band) and to SciPy's reverse Cuthill-McKee (grid); deterministic; row_order
    keeps the natural order for dense or small patterns."""
    import scipy.sparse as sps
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    from oracle import lp_solver as L
    rng = np.random.default_rng(3)
    n = 400
    hid = rng.permutation(n)                              # row i of the hidden banded matrix is stored as row hid[i]
    cols_of = [None] * n
    for i in range(n):
        cols_of[hid[i]] = np.unique(np.clip(np.arange(i - 3, i + 4), 0, n - 1))      # shares columns with hidden neighbours within 6
    order, bw = L.rcm_order(cols_of)
    assert sorted(order.tolist()) == list(range(n))
    pos = np.empty(n, int); pos[order] = np.arange(n)
    real = max(abs(pos[a] - pos[b]) for a in range(n) for b in range(n) if len(np.intersect1d(cols_of[a], cols_of[b])) > 0)
    assert real == bw and bw <= 12
    order2, bw2 = L.rcm_order(cols_of)
    assert np.array_equal(order, order2) and bw == bw2
    g = 20                                                 # rows = edges of a 20 x 20 grid graph, columns = its vertices
    edges = [(r * g + c, r * g + c + 1) for r in range(g) for c in range(g - 1)] + [(r * g + c, (r + 1) * g + c) for r in range(g - 1) for c in range(g)]
    rows_cols = [np.array(e) for e in edges]
    order, bw = L.rcm_order(rows_cols)
    A = sps.csr_matrix((np.ones(2 * len(edges)), (np.repeat(np.arange(len(edges)), 2), np.array(edges).ravel())))
    S = (A @ A.T).tocsr()
    ps = reverse_cuthill_mckee(S, symmetric_mode=True)
    c = S[ps][:, ps].tocoo()
    assert bw <= 1.5 * np.abs(c.row - c.col).max() and bw < len(edges) // 4
    assert L.row_order(rows_cols[:100], g * g) is None                                  # fewer than ROW_ORDER_MIN_M rows
    assert L.row_order([np.arange(50)] * 300, 50) is None                               # dense pattern
    rp = L.row_order(rows_cols, g * g)
    assert rp is not None and np.array_equal(np.argsort(rp), order)


def test_best_iterate_snapshot_and_last_resort_rule(monkeypatch):
    """The best-iterate safeguard of solve_scaled (mirrored in Solver::solve_scaled_impl): (1) IPM.snapshot / IPM.restore bring back an
    iterate exactly (the three measures are recomputed bit for bit); (2) with every active-set attempt made to fail, the solver ends on
    the last resort: the converged iterate is the answer ('ipm-conv', status OPTIMAL, counted as non-canonical) and it is the point the
    interior-point iteration converged to - feasible for the LP to the acceptance IPM_ACCEPT."""
    from oracle import lp_solver as L
    from tests.util import random_subproblem, oracle_solve
    sp = random_subproblem(5, 60, 40, 0.3, 0.0, 2)
    qp, ref = oracle_solve(sp)
    assert ref[5] == 1
    slp, c, rho, kap = L.scale_lp(qp.build_lp(sp['x_k'], sp['delta'], False))
    ip = L.IPM(slp, None, None)
    ip.ns_ok = False
    ip.run(1e-4, 60)
    m0 = ip.measures()
    snap = ip.snapshot()
    ip.run(1e-10, 60)
    assert max(ip.measures()) < max(m0)
    ip.restore(snap)
    assert ip.measures() == m0
    # every polish attempt fails -> last resort
    monkeypatch.setattr(L, "eqp_loop", lambda *a, **k: (False, None, None, None, None))
    monkeypatch.setattr(L, "face_polish", lambda *a, **k: (None, None, None, None, None))
    qp2, out = oracle_solve(sp)
    assert out[5] == 1 and out[6]['stats']['path'] == 'ipm-conv'
    assert np.abs(out[0] - ref[0]).max() <= 1e-6 * max(1.0, np.abs(ref[0]).max())
