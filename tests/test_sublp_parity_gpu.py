"""HIP sub-problem solve vs the CPU oracle on the same seeded inputs, through the C ABI
(asm_sublp_setup / asm_sublp_solve).  Bars: status and active sets identical; step, multipliers and
slack values within 1e-10 relative (BASELINE.json north_star)."""
import numpy as np
import pytest

from tests.util import random_subproblem, equality_rich_subproblem, oracle_solve, hip_solve, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _compare(o_out, h_out, opt, qp_info, sets=None):
    """`sets`: the handle's active_set() taken right after the call (the handle only keeps the last call's)."""
    Xo, lo, uo, Lo, pso, sto, info = o_out
    Xh, lh, uh, Lh, psh, sth = h_out
    assert sth == sto
    if sto != 1:
        assert not Xh.any() and not lh.any()
        return
    rows, bnd, sl = sets if sets is not None else opt.active_set()
    orow, obst, osst = info['sets']
    assert np.array_equal(rows, orow) and np.array_equal(bnd, obst) and np.array_equal(sl, osst)
    assert rel_err(Xh, Xo) < TOL
    assert rel_err(lh, lo) < TOL
    assert rel_err(uh, uo) < TOL and rel_err(Lh, Lo) < TOL
    for i in pso:
        assert rel_err(psh[i], pso[i]) < TOL


@pytest.mark.parametrize("seed,n,m,density,dup,nrange", [(11, 8, 5, 1.0, 0.0, 0), (12, 40, 25, 0.3, 0.3, 3), (13, 120, 60, 1.0, 0.0, 0),
                                                         (14, 200, 150, 0.2, 0.1, 6), (15, 300, 100, 1.0, 0.0, 4)])
def test_normal_phase(seed, n, m, density, dup, nrange):
    sp = random_subproblem(seed, n, m, density, dup, nrange)
    qp, o_out = oracle_solve(sp)
    opt, h_out = hip_solve(sp)
    _compare(o_out, h_out, opt, None)
    st = opt.last_stats()
    assert st['polished'] == 1
    opt.close()


@pytest.mark.parametrize("seed,n,m,nrange", [(21, 10, 8, 2), (22, 60, 40, 4), (23, 150, 90, 0)])
def test_infeasible_then_restoration(seed, n, m, nrange):
    """INFEASIBLE LP -> all-zero outputs (subproblem.jl:532-536), then the feasibility-restoration LP with
    the literal `b -= abs(viol)` shift and slack bounds (subproblem.jl:287-381)."""
    sp = random_subproblem(seed, n, m, 0.5, 0.2, nrange, infeasible=True)
    qp, o_out = oracle_solve(sp)
    opt, h_out = hip_solve(sp)
    assert o_out[5] == 2 and h_out[5] == 2
    _compare(o_out, h_out, opt, None)
    qp, o_out = oracle_solve(sp, True, qp)
    opt, h_out = hip_solve(sp, True, opt)
    assert o_out[5] == 1
    _compare(o_out, h_out, opt, None)
    opt.close()


def test_warm_sequence_matches():
    """A sequence of calls on one handle (the retained active set plays GLPK's retained basis,
    slp.jl:38-40): same path decisions and results as the oracle at every call."""
    sp = random_subproblem(31, 80, 50, 1.0, 0.0, 2)
    qp = opt = None
    rng = np.random.default_rng(5)
    for call in range(5):
        sp2 = dict(sp)
        sp2['dE'] = sp['dE'] * (1.0 + 1e-3 * call * rng.standard_normal(len(sp['dE'])))
        sp2['delta'] = sp['delta'] * (1.0 if call < 3 else 0.5)
        qp, o_out = oracle_solve(sp2, False, qp)
        opt, h_out = hip_solve(sp2, False, opt)
        _compare(o_out, h_out, opt, None)
        assert (opt.last_stats()['path'] == 0) == (o_out[6]['stats']['path'] == 'warm')
    opt.close()


PATH_NAMES = {0: 'warm', 1: 'ipm0+ln', 2: 'ipm1+ln', 3: 'ipm2+ln', 4: 'ipm+face', 5: 'ipm-unpolished', 6: 'ipm-infeasible',
              7: 'phase1-infeasible', 8: 'ipm~+ln', 9: 'ipm+ref', 10: 'ipm-conv'}


@pytest.mark.parametrize("seed,n,m,density,dup,nrange", [(41, 600, 380, 0.012, 0.2, 5), (42, 900, 700, 0.006, 0.0, 8)])
def test_sparse_pattern_parity(seed, n, m, density, dup, nrange):
    """Fill below 1/16: the matrix-vector products run on the CSR/CSC copy of the pattern and the Schur builds skip
    empty k-chunks; results must not differ from the oracle's dense arithmetic beyond the usual bar."""
    sp = random_subproblem(seed, n, m, density, dup, nrange)
    qp, o_out = oracle_solve(sp)
    opt, h_out = hip_solve(sp)
    _compare(o_out, h_out, opt, None)
    assert PATH_NAMES[opt.last_stats()['path']] == o_out[6]['stats']['path']
    opt.close()


def test_restoration_sequence_path_decisions_match():
    """A run of restoration-phase LPs on one handle: the per-phase hints (skipped warm attempts after failures, the
    polish-from-the-interior-iterate preference for non-unique optima) are state carried from call to call; the
    HIP solver must take the oracle's path at every call and return the same point."""
    sp = random_subproblem(51, 120, 90, 0.15, 0.1, 4, infeasible=True)
    qp, o_out = oracle_solve(sp)
    opt, h_out = hip_solve(sp)
    assert o_out[5] == 2 and h_out[5] == 2
    rng = np.random.default_rng(9)
    paths = []
    for call in range(6):
        sp2 = dict(sp)
        sp2['dE'] = sp['dE'] * (1.0 + 2e-2 * rng.standard_normal(len(sp['dE'])))
        sp2['E'] = sp['E'] + 1e-2 * rng.standard_normal(len(sp['E']))
        sp2['x_k'] = np.clip(sp['x_k'] + 0.05 * rng.standard_normal(len(sp['x_k'])), -0.9, 0.9)
        qp, o_out = oracle_solve(sp2, True, qp)
        opt, h_out = hip_solve(sp2, True, opt)
        assert o_out[5] == h_out[5]
        st = opt.last_stats()
        assert PATH_NAMES[st['path']] == o_out[6]['stats']['path'], (call, st, o_out[6]['stats'])
        assert st['eqp'] == o_out[6]['stats']['eqp'] and st['ipm_iters'] == o_out[6]['stats']['ipm_iters']
        if o_out[5] == 1 and o_out[6]['stats']['polished'] == 1:
            _compare(o_out, h_out, opt, None)
        paths.append(o_out[6]['stats']['path'])
    opt.close()


def test_random_campaign():
    """Seeded random sub-problems of mixed size, density, duplicates, range rows, radius, with the restoration LP after an
    INFEASIBLE outcome and a perturbed re-solve on the same handle (warm path): status, path decision, active sets and
    the 1e-10 bar on every call - LPs with a non-unique optimum included (path 'ipm+face': least-norm point of the optimal
    face).  Instances with more equality rows than variables are left out: their active rows are linearly dependent and
    which dependent row the pivot guard drops is a rounding-level decision."""
    checked = 0
    for k in range(60):
        seed = 7000 + k
        rng = np.random.default_rng(seed)
        n = int(rng.integers(4, 200)); m = int(rng.integers(2, 150))
        if m // 3 >= n:
            continue
        dens = float(rng.choice([1.0, 0.5, 0.1, 0.03])); dup = float(rng.choice([0.0, 0.2])); nr = int(rng.integers(0, min(m, 6)))
        infeas = bool(rng.random() < 0.3); delta = float(rng.choice([0.4, 0.05, 1000.0]))
        sp = random_subproblem(seed, n, m, dens, dup, nr, infeasible=infeas, delta=delta)
        def snap():                          # the handle only keeps the last call's sets and statistics
            return (opt.active_set() if h_out[5] == 1 else None), opt.last_stats()
        qp, o_out = oracle_solve(sp)
        opt, h_out = hip_solve(sp)
        calls = [(o_out, h_out) + snap()]
        if o_out[5] == 2:
            qp, o_out = oracle_solve(sp, True, qp)
            opt, h_out = hip_solve(sp, True, opt)
            calls.append((o_out, h_out) + snap())
        sp2 = dict(sp); sp2['dE'] = sp['dE'] * (1.0 + 1e-3 * rng.standard_normal(len(sp['dE'])))
        qp, o_out = oracle_solve(sp2, False, qp)
        opt, h_out = hip_solve(sp2, False, opt)
        calls.append((o_out, h_out) + snap())
        # EVERY call of the seed: status, path, working sets, values (round 2 compared the last call only)
        for oo, hh, sets, st in calls:
            so = oo[6]['stats']
            assert oo[5] == hh[5], seed
            assert PATH_NAMES[st['path']] == so['path'], (seed, st, so)
            if oo[5] == 1:                   # every OPTIMAL answer is an active-set solve on discrete sets ('ipm+ref', the projection
                _compare(oo, hh, opt, None, sets)      # of the interior iterate, does not occur in this campaign: asserted here)
                assert so['path'] != 'ipm+ref' and so['polished'] == 1, (seed, so)
        checked += 1
        opt.close()
    assert checked >= 40


def test_edge_case_shapes():
    """No rows, one variable, rows without entries, all variables fixed, zero radius, cancelling duplicates: the HIP path
    must return the hand-worked answers (and the oracle's) without special-casing by the caller."""
    from tests.util import edge_case_subproblems, EDGE_CASE_ANSWERS
    for name, sp in edge_case_subproblems().items():
        qp, o_out = oracle_solve(sp)
        opt, h_out = hip_solve(sp)
        assert h_out[5] == 1, name
        p_ref, lam_ref = EDGE_CASE_ANSWERS[name]
        assert np.allclose(h_out[0], p_ref, atol=1e-12) and np.allclose(h_out[1], lam_ref, atol=1e-12), name
        _compare(o_out, h_out, opt, None)
        opt.close()


def test_c_abi_argument_errors(hip_lib):
    import ctypes as C
    from activesetmethods_amd.subproblem import QpData, HipSubOptimizer, AsmHipError
    sp = random_subproblem(1, 5, 3)
    free = sp['c_lb'].copy(); free[0] = -np.inf
    ub = sp['c_ub'].copy(); ub[0] = np.inf
    with pytest.raises(AsmHipError):
        HipSubOptimizer(QpData(sp['df'], 0.0, sp['dE'], sp['E'], free, ub, sp['v_lb'], sp['v_ub']), sp['j_row'], sp['j_col'])
    bad = sp['j_row'].copy(); bad[0] = 99
    with pytest.raises(AsmHipError):
        HipSubOptimizer(QpData(sp['df'], 0.0, sp['dE'], sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub']), bad, sp['j_col'])
    h = C.c_void_p()
    assert hip_lib.asm_create(0, C.byref(h)) == 0
    st = C.c_int32(0)
    x = np.zeros(5)
    p = x.ctypes.data_as(C.POINTER(C.c_double))
    assert hip_lib.asm_sublp_solve_resident(h, 1.0, 0, p, p, p, p, p, C.byref(st)) == -3      # ASM_ERR_STATE: no setup
    hip_lib.asm_destroy(h)


def _lp_properties(sp, out, opt, feasibility=False):
    """Size-independent checks of an OPTIMAL sub-LP solution against the LP it solves (normal phase):
    primal feasibility of rows and box, dual sign feasibility, stationarity df - J'lambda - mult = 0 on the
    variables whose own bound (not the trust region) is active or that are free, complementarity, and
    primal objective == dual objective."""
    X, lam, mU, mL, ps, st = out
    assert st == 1
    n, m = sp['n'], sp['m']
    J = np.zeros((m, n))
    np.add.at(J, (sp['j_row'] - 1, sp['j_col'] - 1), sp['dE'])
    lb = np.maximum(-sp['delta'], sp['v_lb'] - sp['x_k']); ub = np.minimum(sp['delta'], sp['v_ub'] - sp['x_k'])
    act = sp['E'] + J @ X
    scale = 1.0 + np.abs(act)
    assert np.all(X >= lb - 1e-12) and np.all(X <= ub + 1e-12)
    assert np.all(act <= sp['c_ub'] + 1e-8 * scale) and np.all(act >= sp['c_lb'] - 1e-8 * scale)
    rows, bnd, sl = opt.active_set()
    assert np.array_equal(X[bnd < 0], lb[bnd < 0]) and np.array_equal(X[bnd > 0], ub[bnd > 0])     # exactly on the bound
    # reduced costs (before the trust-region zeroing of subproblem.jl:522-529)
    z = sp['df'] - J.T @ lam
    zs = max(1.0, np.abs(sp['df']).max())
    free = bnd == 0
    assert np.abs(z[free]).max(initial=0.0) <= 2e-6 * zs
    assert np.all(z[bnd < 0] >= -2e-6 * zs) and np.all(z[bnd > 0] <= 2e-6 * zs)
    ineq_lo = (sp['c_lb'] > -np.inf) & (sp['c_lb'] < sp['c_ub']); ineq_up = (sp['c_ub'] < np.inf) & (sp['c_lb'] < sp['c_ub'])
    only_lo = ineq_lo & ~ineq_up; only_up = ineq_up & ~ineq_lo
    assert np.all(lam[only_lo] >= 0) and np.all(lam[only_up] <= 0)
    # complementarity and strong duality
    gap_rows = np.where(lam > 0, act - sp['c_lb'], np.where(lam < 0, sp['c_ub'] - act, 0.0))
    assert np.abs(gap_rows * lam).max(initial=0.0) <= 1e-7 * zs * scale.max()
    primal = sp['df'] @ X
    r_lo = sp['c_lb'] - sp['E']; r_up = sp['c_ub'] - sp['E']
    dual = np.sum(np.where(lam > 0, lam * np.where(np.isfinite(r_lo), r_lo, 0.0), np.where(lam < 0, lam * np.where(np.isfinite(r_up), r_up, 0.0), 0.0)))
    dual += np.sum(np.where(z > 0, z * lb, z * ub)[~free])
    assert abs(primal - dual) <= 1e-6 * max(1.0, abs(primal))


def test_full_size_c2_lp_properties():
    """BASELINE.json configs[1] at full size (n=1000, m=500 dense): LP optimality certificates recomputed in NumPy."""
    from activesetmethods_amd import problems
    pr = problems.synthetic_dense_nlp(1000, 500)
    x = pr.x0.copy()
    sp = dict(n=pr.n, m=pr.m, j_row=pr.j_row, j_col=pr.j_col, dE=pr.eval_jac_g(x, np.zeros(pr.nnz)), df=pr.eval_grad_f(x, np.zeros(pr.n)),
              f=pr.eval_f(x), E=pr.eval_g(x, np.zeros(pr.m)), x_k=x, c_lb=pr.g_L, c_ub=pr.g_U, v_lb=pr.x_L, v_ub=pr.x_U, delta=0.4)
    opt, out = hip_solve(sp)
    _lp_properties(sp, out, opt)
    assert opt.last_stats()['polished'] == 1
    opt.close()


def test_full_size_c4_lp_properties():
    """BASELINE.json configs[3] size (case1354pegase-sized synthetic ACOPF, 18637 x 11192): the oracle cannot run
    this size in seconds, so the solution is checked through the LP's own optimality certificates."""
    from activesetmethods_amd import acopf
    pr = acopf.acopf_problem(acopf.synthetic_case("case1354pegase", 1), "c4")
    x = pr.x0.copy()
    sp = dict(n=pr.n, m=pr.m, j_row=pr.j_row, j_col=pr.j_col, dE=pr.eval_jac_g(x, np.zeros(pr.nnz)), df=pr.eval_grad_f(x, np.zeros(pr.n)),
              f=pr.eval_f(x), E=pr.eval_g(x, np.zeros(pr.m)), x_k=x, c_lb=pr.g_L, c_ub=pr.g_U, v_lb=pr.x_L, v_ub=pr.x_U, delta=1000.0)
    opt, out = hip_solve(sp)
    _lp_properties(sp, out, opt)
    opt.close()


def _restoration_lp_properties(sp, out, ftol=1e-8, vtol=1e-7):
    """Optimality certificates of a restoration-phase solution, recomputed in NumPy from the LP that subproblem.jl:250-382
    poses: rows with the literal `b -= abs(viol)` shift and their +-unit slack columns, slack lower bounds, objective
    sum(slacks).  Checks primal feasibility, dual sign feasibility, and the optimal value against HiGHS (the value is
    unique also when the optimal point is not)."""
    X, lam, mU, mL, ps, st = out
    assert st == 1
    n, m = sp['n'], sp['m']
    J = np.zeros((m, n))
    np.add.at(J, (sp['j_row'] - 1, sp['j_col'] - 1), sp['dE'])
    c_lb, c_ub, b = sp['c_lb'], sp['c_ub'], sp['E']
    viol = np.where(b > c_ub, c_ub - b, np.where(b < c_lb, c_lb - b, 0.0))
    bs = b - np.abs(viol)
    lb = np.maximum(-sp['delta'], sp['v_lb'] - sp['x_k']); ub = np.minimum(sp['delta'], sp['v_ub'] - sp['x_k'])
    assert np.all(X >= lb - 1e-12) and np.all(X <= ub + 1e-12)
    Jp = J @ X
    two = np.isfinite(c_lb) & np.isfinite(c_ub)
    eq = c_lb == c_ub
    s1 = np.array([ps[i][0] for i in range(m)]); s2 = np.array([ps[i][1] if len(ps[i]) > 1 else 0.0 for i in range(m)])
    slo1 = np.where(two, np.where(viol < 0, 0.0, -viol), -np.abs(viol))
    slo2 = np.where(two, np.where(viol < 0, viol, 0.0), 0.0)
    tol = ftol * (1.0 + np.abs(bs) + np.abs(Jp))
    assert np.all(s1 >= slo1 - tol) and np.all(s2[two] >= slo2[two] - tol[two])
    r_lo = c_lb - bs; r_up = c_ub - bs
    only_lo = np.isfinite(c_lb) & ~np.isfinite(c_ub); only_up = np.isfinite(c_ub) & ~np.isfinite(c_lb); rng = two & ~eq
    assert np.all(np.abs(s1 - s2 + Jp - r_lo)[eq] <= tol[eq]), float((np.abs(s1 - s2 + Jp - r_lo)[eq] / tol[eq]).max() * ftol)
    assert np.all((s1 + Jp - r_lo)[rng | only_lo] >= -tol[rng | only_lo])
    assert np.all((-s2 + Jp - r_up)[rng] <= tol[rng])
    assert np.all((-s1 + Jp - r_up)[only_up] <= tol[only_up])
    assert np.all(lam[only_lo] >= 0) and np.all(lam[only_up] <= 0)
    # optimal value against an independent LP code on the identical LP (the merged multiplier of a range row,
    # subproblem.jl:513-515, does not determine the duals of its two LP rows, so strong duality cannot be recomputed
    # from the outputs): SciPy's HiGHS on the sparse form with explicit slack columns
    import scipy.sparse as sps
    from scipy.optimize import linprog
    from oracle.subproblem import QpData, QpModel, compute_jacobian_matrix
    A0, stored = compute_jacobian_matrix(m, n, sp['j_row'] - 1, sp['j_col'] - 1, sp['dE'])
    qp = QpModel(QpData(sp['df'], sp['f'], A0, sp['E'], c_lb, c_ub, sp['v_lb'], sp['v_ub'], stored), sp['j_row'], sp['j_col'])
    lp = qp.build_lp(sp['x_k'], sp['delta'], True)
    Afull = sps.hstack([sps.csr_matrix(lp.A), sps.csr_matrix((lp.scoef, (lp.srow, np.arange(lp.ns))), shape=(lp.M, lp.ns))]).tocsr()
    e, g, l = lp.rtype == 0, lp.rtype == 1, lp.rtype == -1
    res = linprog(np.concatenate([lp.q, lp.w]), A_ub=sps.vstack([-Afull[g], Afull[l]]).tocsr(), b_ub=np.concatenate([-lp.r[g], lp.r[l]]),
                  A_eq=Afull[e], b_eq=lp.r[e], bounds=np.r_[np.c_[lp.lb, lp.ub], np.c_[lp.slo, np.full(lp.ns, np.inf)]], method="highs")
    assert res.status == 0
    primal = s1.sum() + s2[two].sum()
    assert abs(primal - res.fun) <= vtol * max(1.0, abs(res.fun)), (primal, res.fun)


@pytest.mark.parametrize("name,delta", [("case118", 0.4), ("case1354pegase", 1000.0)])
def test_full_size_restoration_lp_properties(name, delta):
    """Restoration-phase LP at the flat start of the case118- and case1354pegase-sized grids (the latter is
    BASELINE.json configs[3] size): checked through the LP's own optimality certificates."""
    from activesetmethods_amd import acopf
    pr = acopf.acopf_problem(acopf.synthetic_case(name, 1), name)
    x = pr.x0.copy()
    sp = dict(n=pr.n, m=pr.m, j_row=pr.j_row, j_col=pr.j_col, dE=pr.eval_jac_g(x, np.zeros(pr.nnz)), df=pr.eval_grad_f(x, np.zeros(pr.n)),
              f=pr.eval_f(x), E=pr.eval_g(x, np.zeros(pr.m)), x_k=x, c_lb=pr.g_L, c_ub=pr.g_U, v_lb=pr.x_L, v_ub=pr.x_U, delta=delta)
    opt, out = hip_solve(sp, True)
    _restoration_lp_properties(sp, out)
    opt.close()


@pytest.mark.parametrize("seed,infeasible", [(61, False), (62, True)])
def test_lp_entry_for_an_moi_optimizer(seed, infeasible):
    """asm_lp_solve takes the LP the way an MOI `external_optimizer` gets it from the unmodified reference (objective, row
    right-hand sides incl. the extra rows of range constraints, column box, slack weights and lower bounds -
    subproblem.jl:250-484) - here produced by the oracle's build_lp.  It must return what asm_sublp_solve returns for the
    same sub-problem (the formulation done inside the library), in both phases."""
    from activesetmethods_amd.subproblem import QpData, HipSubOptimizer
    sp = random_subproblem(seed, 50, 36, 0.4, 0.2, 4, infeasible=infeasible)
    qp, o_out = oracle_solve(sp)
    opt, h_out = hip_solve(sp)
    fr = o_out[5] == 2
    if fr:
        qp, o_out = oracle_solve(sp, True, qp)
        opt, h_out = hip_solve(sp, True, opt)
    assert o_out[5] == h_out[5] == 1
    lp = qp.build_lp(sp['x_k'], sp['delta'], fr)
    opt2 = HipSubOptimizer(QpData(sp['df'], sp['f'], sp['dE'], sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub']), sp['j_row'], sp['j_col'])
    p, s, y, z, bs, st = opt2.lp_solve(sp['dE'], lp.q, lp.r, lp.lb, lp.ub, lp.w if fr else None, lp.slo if fr else None)
    assert st == 1
    assert rel_err(p, h_out[0]) < 1e-12
    m = sp['m']
    lam = y[:m].copy()
    for k, val in enumerate(qp.adj):
        lam[val] += y[m + k]
    assert rel_err(lam, h_out[1]) < 1e-12
    if fr:
        flat = np.concatenate([np.asarray(h_out[4][i], float) for i in range(m)])
        assert rel_err(s, flat) < 1e-12
    assert np.array_equal(bs, opt.active_set()[1])
    opt.close(); opt2.close()


@pytest.mark.parametrize("seed,infeasible", [(71, False), (72, True)])
def test_scalar_readback_paths_agree(seed, infeasible, monkeypatch):
    """The interior-point loop reads its scalar block either from host-mapped memory the reduction kernels write themselves (default:
    the host spins on a sequence word) or with hipMemcpyAsync + hipStreamSynchronize (ASM_HIP_SPIN=0).  Same kernels, same values:
    the two paths must give bit-identical answers and statistics."""
    sp = random_subproblem(seed, 60, 45, 0.4, 0.2, 4, infeasible=infeasible)
    outs = []
    for spin in ("1", "0"):
        monkeypatch.setenv("ASM_HIP_SPIN", spin)           # read at asm_create
        opt, h_out = hip_solve(sp)
        if h_out[5] == 2:
            opt, h_out = hip_solve(sp, True, opt)
        st = opt.last_stats()
        outs.append((h_out, st['ipm_iters'], st['nfact'], st['path']))
        opt.close()
    (a, ia, fa, pa), (b, ib, fb, pb) = outs
    assert (ia, fa, pa) == (ib, fb, pb)
    assert a[5] == b[5] == 1
    for u, v in zip(a[:4], b[:4]):
        assert np.array_equal(u, v)


# ----------------------------------------------------------------------------- round 3
@pytest.mark.parametrize("seed,n,neq,nineq", [(81, 300, 260, 160), (82, 500, 470, 200), (83, 200, 150, 260)])
def test_null_space_form_parity(seed, n, neq, nineq):
    """Equality-rich sparse sub-problems (the ACOPF row structure): the interior-point iterations factor the k x k null-space form, the
    active-set solves run in reduced coordinates, the basis is carried from LP to LP - three LPs on one handle, each compared with
    the oracle: status, path, working sets, iteration counts, 1e-10 on step and multipliers."""
    sp = equality_rich_subproblem(seed, n, neq, nineq)
    rng = np.random.default_rng(seed)
    qp = opt = None
    cold = []
    for call in range(3):
        sp2 = dict(sp)
        if call:
            sp2['dE'] = sp['dE'] * (1.0 + 2e-2 * rng.standard_normal(len(sp['dE'])))
            sp2['df'] = sp['df'] + 0.1 * rng.standard_normal(n)
        qp, o_out = oracle_solve(sp2, False, qp)
        opt, h_out = hip_solve(sp2, False, opt)
        st, so = opt.last_stats(), o_out[6]['stats']
        assert o_out[5] == h_out[5] == 1, (call, o_out[5], h_out[5])
        assert st['ns_dim'] == n - neq and st['ns_iters'] > 0, st                       # the form was used (k = n - #equality rows)
        assert st['ns_iters'] == so['ns_iters'] and st['ipm_iters'] == so['ipm_iters'], (call, st, so)
        assert PATH_NAMES[st['path']] == so['path'], (call, st, so)
        _compare(o_out, h_out, opt, None)
        cold.append(st['ns_cold'])
    assert cold == [1, 0, 0], cold                                                      # basis selected once, then carried
    opt.close()


def _acopf_subproblem(pr, x, delta):
    return dict(n=pr.n, m=pr.m, j_row=pr.j_row, j_col=pr.j_col, dE=pr.eval_jac_g(x, np.zeros(pr.nnz)), df=pr.eval_grad_f(x, np.zeros(pr.n)),
                f=pr.eval_f(x), E=pr.eval_g(x, np.zeros(pr.m)), x_k=x.copy(), c_lb=pr.g_L, c_ub=pr.g_U, v_lb=pr.x_L, v_ub=pr.x_U, delta=delta)


def _highs_value(pr, sp, fr):
    """Optimal value of the sub-LP from an independent code (SciPy's HiGHS on the sparse statement, oracle/sparse_lp.py)."""
    from oracle import sparse_lp
    lp = sparse_lp.build(pr.n, pr.m, pr.j_row, pr.j_col, sp['dE'], sp['df'], sp['E'], pr.g_L, pr.g_U, pr.x_L, pr.x_U, sp['x_k'], sp['delta'], fr)
    st, obj, p, dt, nit = sparse_lp.solve_highs(lp)
    return st, obj


def test_c4_size_three_consecutive_normal_phase_lps():
    """case1354pegase-sized grid at load scale 0.5 (the bench workload): three consecutive normal-phase LPs on ONE handle along the
    SLP's own steps (the second and third re-use the null-space basis), each checked through the LP's optimality certificates
    recomputed in NumPy and its optimal value against HiGHS."""
    from activesetmethods_amd import acopf
    pr = acopf.acopf_problem(acopf.synthetic_case("case1354pegase", 1, 0.5), "c4")
    x = pr.x0.copy()
    opt = None
    for k in range(3):
        sp = _acopf_subproblem(pr, x, 1000.0)
        opt, out = hip_solve(sp, False, opt)
        assert out[5] == 1
        st = opt.last_stats()
        assert st['polished'] == 1 and st['ns_iters'] > 0 and st['ns_cold'] == (1 if k == 0 else 0), st
        _lp_properties(sp, out, opt)
        hs, hobj = _highs_value(pr, sp, False)
        assert hs == 1
        assert abs(sp['df'] @ out[0] - hobj) <= 1e-7 * max(1.0, abs(hobj)), (k, sp['df'] @ out[0], hobj)
        x = np.clip(x + 0.1 * out[0], pr.x_L, pr.x_U)
    opt.close()


def test_c4_size_three_consecutive_restoration_lps():
    """The nominal-load case1354pegase-sized grid (never leaves feasibility restoration): three consecutive restoration LPs on ONE
    handle along damped steps, each with the NumPy certificates and the HiGHS optimal value of _restoration_lp_properties."""
    from activesetmethods_amd import acopf
    pr = acopf.acopf_problem(acopf.synthetic_case("case1354pegase", 1), "c4fr")
    x = pr.x0.copy()
    opt = None
    for k in range(3):
        sp = _acopf_subproblem(pr, x, 1000.0)
        opt, out = hip_solve(sp, True, opt)
        assert opt.last_stats()['polished'] == 1
        _restoration_lp_properties(sp, out)
        x = np.clip(x + 0.05 * out[0], pr.x_L, pr.x_U)
    opt.close()


def test_restoration_run_at_load_0_8_has_no_unpolished_lp():
    """The case1354pegase-sized grid at load scale 0.8: round 3 / early round 4 stopped this Line-Search run with status -5 after 8 LPs because a
    restoration LP came back unpolished (path 5): the column-form interior-point iterate stalls at a dual residual of 2e-9 ... 7e-9 on a
    degenerate optimal face, no active-set solve confirms a partition, and the late iterations degrade the primal residual.  Now the best
    stage end is kept (best-iterate safeguard) and an iterate converged to 1e-8 (GLPK's own tolerances are 1e-7) is the last-resort
    answer ('ipm-conv', path 10, counted as non-canonical): every LP of the first 12 ends OPTIMAL, and every last-resort answer passes the
    restoration LP's own certificates and agrees with HiGHS on the optimal value (1e-7 relative)."""
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf
    pr = acopf.function_model(acopf.synthetic_case("case1354pegase", 1, 0.8)).to_problem("load 0.8")      # device evaluator, as in the benchmark
    slp = A.optimize(A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=100, device_eval=True)), max_lp_solves=12)
    slp.optimizer.close()
    assert len(slp.trace) == 12
    # (status 2 = the normal-phase LP is infeasible: the run enters feasibility restoration, slp_line_search.jl:135-142)
    assert all(r['status'] in (1, 2) and r['stats']['path'] != 5 for r in slp.trace), [(r['status'], r['stats']['path']) for r in slp.trace]
    last_resort = [r for r in slp.trace if r['stats']['path'] == 10]
    assert len(last_resort) >= 1 and all(r['fr'] for r in last_resort), [r['stats']['path'] for r in slp.trace]
    for rec in last_resort[:2]:
        assert max(rec['stats']['ipm_pinf'], rec['stats']['ipm_dinf'], rec['stats']['ipm_gap']) <= 1e-8
        sp = _acopf_subproblem(pr, rec['x'], rec['delta'])
        # (an interior iterate, not an active-set answer at 1e-13: measured 1.05e-6 relative on the equality rows in the caller's units for a
        # scaled primal residual of 4.5e-11 - the weakest path gets 5e-6 here; optimal value 1e-6 as in test_converged_iterate_last_resort_scenario_27)
        _restoration_lp_properties(sp, (rec['p'], rec['lam'], rec['mult_x_U'], rec['mult_x_L'], rec['p_slack'], 1), ftol=5e-6, vtol=1e-6)


def test_null_space_form_edge_paths():
    """Sequences on one handle that leave the common path of the null-space form: variables that become fixed between LPs (the
    null-space dimension changes: the carried basis and the retained columns are dropped and selected afresh), a duplicated
    equality row (S0 singular: the guarded factor drops it, k grows by one), a zero trust region (every variable fixed: the form is
    not used), and back.  Every call against the oracle: status, path, sets, 1e-10."""
    sp = equality_rich_subproblem(88, 300, 250, 170)
    n, m = sp['n'], sp['m']
    rng = np.random.default_rng(88)
    variants = []
    variants.append(dict(sp))                                                   # 0: plain
    a = dict(sp); a['v_lb'] = sp['v_lb'].copy(); a['v_ub'] = sp['v_ub'].copy()
    fix = rng.choice(n, 12, replace=False)
    a['v_lb'][fix] = sp['x_k'][fix]; a['v_ub'][fix] = sp['x_k'][fix]            # 1: twelve variables fixed at the current point
    variants.append(a)
    variants.append(dict(sp))                                                   # 2: free again
    b = dict(sp); b['delta'] = 0.0                                              # 3: zero radius (all fixed)
    variants.append(b)
    c = dict(sp); c['dE'] = sp['dE'] * (1.0 + 1e-2 * rng.standard_normal(len(sp['dE'])))   # 4: perturbed, basis carried again
    variants.append(c)
    qp = opt = None
    dims = []
    for k, v in enumerate(variants):
        if k in (1, 2):                         # bounds are part of the skeleton state of the oracle's QpModel data only; the handle gets them via set_bounds
            pass
        qp, o_out = oracle_solve(v, False, qp)
        if opt is not None and k in (1, 2, 3, 4):
            from activesetmethods_amd.subproblem import QpData
            opt.set_bounds(QpData(v['df'], v['f'], v['dE'], v['E'], v['c_lb'], v['c_ub'], v['v_lb'], v['v_ub']))
            keep = qp.hint[False].get('ns_J')          # set_bounds drops the retained working sets and the basis, it keeps the basis COLUMNS
            qp.warm = {False: None, True: None}; qp.hint = {False: ({} if keep is None else {'ns_J': keep}), True: {'prefer_ref': True}}
        opt, h_out = hip_solve(v, False, opt)
        st, so = opt.last_stats(), o_out[6]['stats']
        assert o_out[5] == h_out[5], (k, o_out[5], h_out[5])
        if o_out[5] == 1:
            assert PATH_NAMES[st['path']] == so['path'], (k, st, so)
            assert st['ns_iters'] == so.get('ns_iters', 0), (k, st, so)
            _compare(o_out, h_out, opt, None)
        dims.append(st['ns_dim'])
    assert dims[0] == n - 250 and dims[1] == n - 12 - 250 and dims[2] == n - 250 and dims[3] == 0 and dims[4] == n - 250, dims
    opt.close()
    # the null space grows beyond what the first LP of the form reserved: 130 of 560 variables fixed at first (k = 30), then free (k = 160) -
    # the k-sized buffers are re-allocated, the form stays in use (round 3 fell back to the M x M row form for good)
    sp3 = equality_rich_subproblem(90, 560, 400, 200)
    fix = np.random.default_rng(90).choice(sp3['n'], 130, replace=False)
    a3 = dict(sp3); a3['v_lb'] = sp3['v_lb'].copy(); a3['v_ub'] = sp3['v_ub'].copy()
    a3['v_lb'][fix] = a3['v_ub'][fix] = (sp3['x_k'] + sp3['p_star'])[fix]          # fixed where the generator's feasible step puts them
    qp = opt = None
    dims = []
    for k, v in enumerate((a3, sp3)):
        qp, o_out = oracle_solve(v, False, qp)
        if opt is not None:
            from activesetmethods_amd.subproblem import QpData
            opt.set_bounds(QpData(v['df'], v['f'], v['dE'], v['E'], v['c_lb'], v['c_ub'], v['v_lb'], v['v_ub']))
        opt, h_out = hip_solve(v, False, opt)
        st, so = opt.last_stats(), o_out[6]['stats']
        assert o_out[5] == h_out[5] == 1 and PATH_NAMES[st['path']] == so['path'] and st['ns_iters'] == so.get('ns_iters', 0) > 0, (k, st, so)
        rows, bnd, sl = opt.active_set()
        assert np.array_equal(rows, o_out[6]['sets'][0]) and np.array_equal(bnd, o_out[6]['sets'][1])
        # (400 random equality rows on 560 columns, 130 of them fixed: a badly conditioned instance - agreement to 1e-8, not the 1e-10 of the campaigns)
        assert rel_err(h_out[0], o_out[0]) < 1e-8 and rel_err(h_out[1], o_out[1]) < 1e-8
        dims.append(st['ns_dim'])
        if k == 0:
            qp.warm = {False: None, True: None}; qp.hint = {False: {'ns_J': qp.hint[False].get('ns_J')}, True: {'prefer_ref': True}}
    assert dims[0] in (30, 31) and dims[1] == 560 - 400, dims      # (with 130 columns fixed one of the 400 random equality rows is dependent: dropped, k + 1)
    opt.close()
    # a duplicated equality row: dependent rows of A_EF
    sp2 = equality_rich_subproblem(89, 260, 200, 150)
    J = sp2['J']
    rows, cols = sp2['j_row'] - 1, sp2['j_col'] - 1
    src = 5; dst = 6                                                            # row 6 := row 5 (same coefficients, same bounds and value)
    keep = rows != dst
    add = rows == src
    j_row = np.concatenate([rows[keep], np.full(add.sum(), dst)]) + 1
    j_col = np.concatenate([cols[keep], cols[add]]) + 1
    dE = np.concatenate([sp2['dE'][keep], sp2['dE'][add]])
    sp3 = dict(sp2); sp3['j_row'] = j_row; sp3['j_col'] = j_col; sp3['dE'] = dE
    sp3['E'] = sp2['E'].copy(); sp3['E'][dst] = sp2['E'][src]
    sp3['c_lb'] = sp2['c_lb'].copy(); sp3['c_ub'] = sp2['c_ub'].copy()
    sp3['c_lb'][dst] = sp2['c_lb'][src]; sp3['c_ub'][dst] = sp2['c_ub'][src]
    qp, o_out = oracle_solve(sp3)
    opt, h_out = hip_solve(sp3)
    st, so = opt.last_stats(), o_out[6]['stats']
    assert o_out[5] == h_out[5] == 1
    assert st['ns_dim'] == sp3['n'] - 200 + 1, st                               # one dependent equality row dropped: the null space has one more dimension
    assert PATH_NAMES[st['path']] == so['path'], (st, so)
    _compare(o_out, h_out, opt, None)
    opt.close()


def test_null_space_campaign():
    """Seeded equality-rich sub-problems of varying shape, with a perturbed re-solve on the same handle: status, path, working
    sets, interior-point and null-space iteration counts, 1e-10 - on every call."""
    checked = 0
    for k in range(12):
        seed = 9100 + k
        rng = np.random.default_rng(seed)
        n = int(rng.integers(150, 500)); neq = int(rng.integers(max(64, n - 120), n - 10)); nineq = int(rng.integers(max(n - neq, 40), 300))
        if n > neq + nineq or n - neq > 0.3 * (neq + nineq):
            continue
        sp = equality_rich_subproblem(seed, n, neq, nineq, per_row=int(rng.integers(3, 6)), delta=float(rng.choice([0.6, 0.2, 1000.0])))
        qp, o1 = oracle_solve(sp)
        opt, h1 = hip_solve(sp)
        s1 = (opt.active_set() if h1[5] == 1 else None), opt.last_stats()
        sp2 = dict(sp); sp2['dE'] = sp['dE'] * (1.0 + 1e-2 * rng.standard_normal(len(sp['dE']))); sp2['df'] = sp['df'] + 0.2 * rng.standard_normal(n)
        qp, o2 = oracle_solve(sp2, False, qp)
        opt, h2 = hip_solve(sp2, False, opt)
        s2 = (opt.active_set() if h2[5] == 1 else None), opt.last_stats()
        for oo, hh, (sets, st) in ((o1, h1, s1), (o2, h2, s2)):
            so = oo[6]['stats']
            assert oo[5] == hh[5], seed
            if oo[5] != 1:
                continue
            assert PATH_NAMES[st['path']] == so['path'], (seed, st, so)
            assert st['ns_iters'] == so['ns_iters'] and st['ipm_iters'] == so['ipm_iters'], (seed, st, so)
            assert st['ns_dim'] > 0
            _compare(oo, hh, opt, None, sets)
        checked += 1
        opt.close()
    assert checked >= 6


def test_row_order_is_the_oracles():
    """The order the factorisations take rows in is part of the answer in degenerate cases (the pivot guard drops the later of two
    dependent rows): the library's host code and the oracle compute the same reverse Cuthill-McKee orders and bandwidths - all rows of the
    internal LP, and the equality rows of the null-space form - on the ACOPF pattern (banded) and on a dense pattern (natural order)."""
    from activesetmethods_amd import acopf
    from activesetmethods_amd.subproblem import HipSubOptimizer, QpData
    from oracle import lp_solver as L, subproblem as OS
    pr = acopf.acopf_problem(acopf.synthetic_case("case300", 1, 0.5), "case300")
    sp = _acopf_subproblem(pr, np.asarray(pr.x0, float).copy(), 1000.0)
    opt = HipSubOptimizer(QpData(sp['df'], sp['f'], sp['dE'], sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub']), sp['j_row'], sp['j_col'])
    perm, band, e_rows, e_band = opt.row_order()
    opt.close()
    J = OS.compute_jacobian_matrix(pr.m, pr.n, np.asarray(sp['j_row']) - 1, np.asarray(sp['j_col']) - 1, sp['dE'])
    qp = OS.QpModel(OS.QpData(sp['df'], sp['f'], J[0], sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub'], J[1]), sp['j_row'], sp['j_col'])
    assert qp.row_pos is not None and perm is not None
    assert np.array_equal(np.argsort(qp.row_pos), perm)
    order_all, bw_all = L.rcm_order(qp.row_cols + [qp.row_cols[v] for v in qp.adj])
    assert band == bw_all and np.array_equal(order_all, perm) and 0 < band < len(perm) // 2
    Erows = np.nonzero(qp.rtype == 0)[0]
    order_e, bw_e = L.rcm_order([qp.row_cols[i] for i in Erows])
    assert np.array_equal(Erows[order_e], e_rows) and e_band == bw_e and 0 < e_band < len(Erows) // 2
    spd = random_subproblem(5, 40, 30)                      # dense pattern, small: natural order on both sides
    opt2, _ = hip_solve(spd)
    assert opt2.row_order()[0] is None and opt2.row_order()[1] == 0
    opt2.close()


@pytest.mark.parametrize("seed,n,m,neq,nrange", [(501, 400, 600, 120, 30), (502, 300, 700, 40, 60), (503, 500, 520, 300, 0), (504, 350, 640, 0, 40)])
def test_banded_row_order_parity(seed, n, m, neq, nrange):
    """Sub-problems whose coupling graph is banded after reordering (tests/util.banded_subproblem, rows stored in random order): the
    library takes row lists in reverse Cuthill-McKee order and factors banded matrices - active-set solves, interior-point row forms
    (full and reduced), the column form of the restoration phase, the null-space form's S0 where it applies.  Normal phase, a perturbed
    re-solve on the same handle, then (infeasible variant) the infeasibility verdict and the restoration LP: status, path, working
    sets, 1e-10 against the oracle on every call; the order in use is the oracle's."""
    from tests.util import banded_subproblem
    sp = banded_subproblem(seed, n, m, neq, nrange)
    qp, o1 = oracle_solve(sp)
    opt, h1 = hip_solve(sp)
    perm, band, _, _ = opt.row_order()
    assert perm is not None and 0 < band < m // 2 and qp.row_pos is not None and np.array_equal(np.argsort(qp.row_pos), perm)
    st = opt.last_stats()
    assert o1[5] == h1[5] == 1 and PATH_NAMES[st['path']] == o1[6]['stats']['path'], (st, o1[6]['stats'])
    _compare(o1, h1, opt, None)
    rng = np.random.default_rng(seed + 1)
    sp2 = dict(sp); sp2['dE'] = sp['dE'] * (1.0 + 1e-2 * rng.standard_normal(len(sp['dE']))); sp2['df'] = sp['df'] + 0.2 * rng.standard_normal(n)
    qp, o2 = oracle_solve(sp2, False, qp)
    opt, h2 = hip_solve(sp2, False, opt)
    st = opt.last_stats()
    assert o2[5] == h2[5] == 1 and PATH_NAMES[st['path']] == o2[6]['stats']['path'], (st, o2[6]['stats'])
    _compare(o2, h2, opt, None)
    opt.close()
    spi = banded_subproblem(seed, n, m, neq, nrange, infeasible=True)
    qp, oi = oracle_solve(spi)
    opt, hi = hip_solve(spi)
    assert oi[5] == hi[5] == 2                                   # INFEASIBLE on both sides
    qp, orr = oracle_solve(spi, True, qp)
    opt, hr = hip_solve(spi, True, opt)
    st = opt.last_stats()
    assert orr[5] == hr[5] == 1 and PATH_NAMES[st['path']] == orr[6]['stats']['path'], (st, orr[6]['stats'])
    assert st['ipm_iters'] == orr[6]['stats']['ipm_iters'] and st['col_iters'] == orr[6]['stats'].get('col_iters', 0), (st, orr[6]['stats'])
    _compare(orr, hr, opt, None)
    opt.close()
