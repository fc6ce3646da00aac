"""End-to-end SLP runs: the HIP path (activesetmethods_amd.slp -> libasmhip) against the oracle's
restatement of the same callers, plus the reference's own known answers for the path
(test/runtests.jl:11-13: toy -> X = Y = -1, LOCALLY_SOLVED)."""
import numpy as np
import pytest

from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _oracle_run(pr, **kw):
    from oracle import slp as O
    m = O.Model(pr.n, pr.m, pr.x_L, pr.x_U, pr.g_L, pr.g_U, pr.j_str, pr.eval_f, pr.eval_g, pr.eval_grad_f, pr.eval_jac_g,
                O.Parameters(**kw))
    m.x[:] = pr.x0
    return m, O.optimize(m)


def _hip_run(pr, **kw):
    import activesetmethods_amd as A
    m = A.Model.from_problem(pr, A.Parameters(**kw))
    return m, A.optimize(m)


def _compare_traces(so, sh, tol=1e-9):
    assert len(so.trace) == len(sh.trace)
    for ro, rh in zip(so.trace, sh.trace):
        assert ro['status'] == rh['status'] and ro['fr'] == rh['fr'] and ro['iter'] == rh['iter']
        if ro['status'] == 1:
            for a, b in zip(ro['sets'], rh['sets']):
                assert np.array_equal(a, b)                       # active-set index sequence identical
            assert rel_err(rh['p'], ro['p']) < tol
            assert rel_err(rh['lam'], ro['lam']) < tol
            assert rel_err(rh['mult_x_U'], ro['mult_x_U']) < tol and rel_err(rh['mult_x_L'], ro['mult_x_L']) < tol


@pytest.mark.parametrize("alg", ["Line Search", "Trust Region"])
def test_toy_known_answer_and_trace(alg):
    from activesetmethods_amd import problems
    mh, sh = _hip_run(problems.toy_problem(), algorithm=alg)
    assert mh.status == 0                                                 # LOCALLY_SOLVED
    assert np.allclose(mh.x, [-1.0, -1.0], rtol=1e-4)                     # test/runtests.jl:11-12
    assert np.allclose(mh.mult_g, [0.0, 1.0 / 3.0, 0.0, 0.0], atol=1e-6)  # from df - J'lambda = 0 (common.jl:38)
    mo, so = _oracle_run(problems.toy_problem(), algorithm=alg)
    assert mo.status == mh.status and so.iter == sh.iter and so.lp_solves == sh.lp_solves
    _compare_traces(so, sh)
    assert rel_err(mh.x, mo.x) < 1e-9


@pytest.mark.parametrize("alg,iters", [("Trust Region", 12), ("Line Search", 8)])
def test_synthetic_dense_small_trace(alg, iters):
    from activesetmethods_amd import problems
    pr = problems.synthetic_dense_nlp(120, 60)
    mh, sh = _hip_run(pr, algorithm=alg, max_iter=iters)
    mo, so = _oracle_run(pr, algorithm=alg, max_iter=iters)
    assert mo.status == mh.status and so.iter == sh.iter
    _compare_traces(so, sh)
    assert rel_err(mh.x, mo.x) < 1e-8
    assert all(r['stats']['polished'] == 1 for r in sh.trace)


def test_full_size_c2_trust_region_trace():
    """BASELINE.json configs[1] at its full size (n = 1000, m = 500, dense Jacobian, 500 000 entries): the first three Trust-Region LPs of the
    SLP run against the oracle's run - same status, phase, working sets (rows / bounds) at every LP, step and multipliers within 1e-9."""
    from activesetmethods_amd import problems
    pr = problems.synthetic_dense_nlp(1000, 500)
    mh, sh = _hip_run(pr, algorithm="Trust Region", max_iter=3)
    mo, so = _oracle_run(pr, algorithm="Trust Region", max_iter=3)
    assert so.lp_solves == sh.lp_solves >= 3 and so.iter == sh.iter
    _compare_traces(so, sh)
    assert rel_err(mh.x, mo.x) < 1e-9
    assert all(r['stats']['polished'] == 1 for r in sh.trace)


@pytest.mark.parametrize("kind", ["dense", "sparse"])
def test_reductions_on_resident_jacobian(kind):
    """KT_residuals / row norms computed from the HBM-resident Jacobian (common.jl:35-44); the sparse case runs the
    products on the CSR/CSC copy of the pattern."""
    from activesetmethods_amd import problems, acopf, QpData, HipSubOptimizer
    from oracle import slp as O
    from oracle.subproblem import compute_jacobian_matrix
    pr = problems.synthetic_dense_nlp(90, 40) if kind == "dense" else acopf.acopf_problem(acopf.synthetic_case("case118", 3), "case118")
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.5, 0.5, pr.n) if kind == "dense" else pr.x0 + 0.01 * rng.standard_normal(pr.n)
    df = pr.eval_grad_f(x, np.zeros(pr.n)); E = pr.eval_g(x, np.zeros(pr.m)); dE = pr.eval_jac_g(x, np.zeros(pr.nnz))
    opt = HipSubOptimizer(QpData(df, 0.0, dE, E, pr.g_L, pr.g_U, pr.x_L, pr.x_U), pr.j_row, pr.j_col)
    opt.upload(dE, df, 0.0, E, x)
    lam = rng.standard_normal(pr.m); mu = rng.standard_normal(pr.n) * 0.1; ml = rng.standard_normal(pr.n) * 0.1
    J, _ = compute_jacobian_matrix(pr.m, pr.n, pr.j_row - 1, pr.j_col - 1, dE)
    ref = O.KT_residuals(df, lam, mu, ml, J)
    assert abs(opt.kt_residuals(df, lam, mu, ml) - ref) < 1e-12 * max(1.0, abs(ref))
    assert rel_err(opt.jac_row_norms(), np.linalg.norm(J, axis=1)) < 1e-13
    opt.close()


def _case3():
    import json, os
    from activesetmethods_amd import acopf
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "case3.json")))
    return acopf.acopf_problem(acopf.case_from_tables(d["baseMVA"], d["bus"], d["gen"], d["gencost"], d["branch"], d["dcline"]), "case3"), d


@pytest.mark.parametrize("alg", ["Line Search", "Trust Region"])
def test_acopf_case3_known_objective(alg):
    """test/runtests.jl:18-19: ACOPF case3 objective 5906.87949 (rtol 1e-3), max_iter 100."""
    pr, d = _case3()
    mh, sh = _hip_run(pr, algorithm=alg, max_iter=100)
    assert mh.status == 0
    assert abs(mh.obj_val - d["expected_objective"]) <= 1e-3 * d["expected_objective"]
    mo, so = _oracle_run(pr, algorithm=alg, max_iter=100)
    assert so.lp_solves == sh.lp_solves
    _compare_traces(so, sh)
    assert abs(mh.obj_val - mo.obj_val) < 1e-6


def test_acopf_case118_sized_lp_parity():
    """One normal-phase sub-LP of the case118-sized synthetic grid (sparse pattern, unbounded angle
    variables, 1725 rows, Line-Search radius 1000)."""
    from activesetmethods_amd import acopf
    from tests.util import oracle_solve, hip_solve
    pr = acopf.acopf_problem(acopf.synthetic_case("case118", 1), "case118")
    x = pr.x0.copy()
    sp = dict(n=pr.n, m=pr.m, j_row=pr.j_row, j_col=pr.j_col, dE=pr.eval_jac_g(x, np.zeros(pr.nnz)), df=pr.eval_grad_f(x, np.zeros(pr.n)),
              f=pr.eval_f(x), E=pr.eval_g(x, np.zeros(pr.m)), x_k=x, c_lb=pr.g_L, c_ub=pr.g_U, v_lb=pr.x_L, v_ub=pr.x_U, delta=1000.0)
    qp, o_out = oracle_solve(sp)
    opt, h_out = hip_solve(sp)
    assert o_out[5] == h_out[5] == 1
    rows, bnd, sl = opt.active_set()
    assert np.array_equal(rows, o_out[6]['sets'][0]) and np.array_equal(bnd, o_out[6]['sets'][1])
    assert rel_err(h_out[0], o_out[0]) < 1e-9 and rel_err(h_out[1], o_out[1]) < 1e-9
    opt.close()


def test_acopf_restoration_lp_with_non_unique_optimum():
    """Trust-Region sub-LP of the case118-sized grid at x0: INFEASIBLE, then the restoration LP `min sum(slacks)`, whose
    optimum is not unique (the optimal face has ~170 dimensions, ~60 of the active rows are linearly dependent).  The solver
    returns the canonical pair - the least-norm point of the optimal face and the basic multipliers - which is determined by
    the LP alone (path 'ipm+face'): same path, counts, working sets as the oracle and the 1e-10 bar on step and multipliers."""
    from activesetmethods_amd import acopf
    from tests.util import oracle_solve, hip_solve
    pr = acopf.acopf_problem(acopf.synthetic_case("case118", 1), "case118")
    x = pr.x0.copy()
    sp = dict(n=pr.n, m=pr.m, j_row=pr.j_row, j_col=pr.j_col, dE=pr.eval_jac_g(x, np.zeros(pr.nnz)), df=pr.eval_grad_f(x, np.zeros(pr.n)),
              f=pr.eval_f(x), E=pr.eval_g(x, np.zeros(pr.m)), x_k=x, c_lb=pr.g_L, c_ub=pr.g_U, v_lb=pr.x_L, v_ub=pr.x_U, delta=0.4)
    qp, o_out = oracle_solve(sp)
    opt, h_out = hip_solve(sp)
    assert o_out[5] == h_out[5] == 2                                       # subproblem.jl:532-536
    for call in range(2):
        qp, o_out = oracle_solve(sp, True, qp)
        opt, h_out = hip_solve(sp, True, opt)
        st, so = opt.last_stats(), o_out[6]['stats']
        assert o_out[5] == h_out[5] == 1
        assert so['path'] == 'ipm+face' and st['path'] == 4
        assert st['ipm_iters'] == so['ipm_iters'] and st['eqp'] == so['eqp']
        rows, bnd, sl = opt.active_set()
        for a, b in zip((rows, bnd, sl), o_out[6]['sets']):
            assert np.array_equal(a, b)
        obj_o = sum(sum(v) for v in o_out[4].values()); obj_h = sum(sum(v) for v in h_out[4].values())
        assert abs(obj_h - obj_o) <= 1e-10 * max(1.0, abs(obj_o))
        for k in range(4):
            assert rel_err(h_out[k], o_out[k]) < 1e-10, k
        assert qp.hint[True].get('prefer_ref') is True
    opt.close()


def test_restoration_sequence_does_not_drift():
    """An SLP run that passes through a sequence of restoration LPs (non-unique optima): because every LP answer is the
    canonical pair, HIP and oracle stay together - per LP identical status / path / working sets and the 1e-9 trace bar, and
    the iterates x agree to 1e-9 after the whole sequence (with the projection of the interior iterate this drifted from
    4e-7 to 6e-2 over ten LPs)."""
    from activesetmethods_amd import acopf
    pr = acopf.acopf_problem(acopf.synthetic_case("case118", 1), "case118")
    mh, sh = _hip_run(pr, algorithm="Trust Region", max_iter=12)
    mo, so = _oracle_run(pr, algorithm="Trust Region", max_iter=12)
    assert len(sh.trace) == len(so.trace) and sum(1 for r in so.trace if r['fr']) >= 10
    names = {0: 'warm', 1: 'ipm0+ln', 2: 'ipm1+ln', 3: 'ipm2+ln', 4: 'ipm+face', 5: 'ipm-unpolished', 6: 'ipm-infeasible', 7: 'phase1-infeasible',
             8: 'ipm~+ln', 9: 'ipm+ref', 10: 'ipm-conv'}
    assert [names[r['stats']['path']] for r in sh.trace] == [r['stats']['path'] for r in so.trace]
    assert all(r['stats']['path'] != 'ipm+ref' for r in so.trace)
    _compare_traces(so, sh)
    assert rel_err(mh.x, mo.x) < 1e-9


def test_scenario_batch_on_one_gpu():
    """Scenario batch path on a single rank: block partition, one handle alive at a time,
    merged statistics.  Scenarios = case3 with per-bus load factors."""
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf, batch
    pr0, d = _case3()
    base = pr0.model.c
    shared = {}

    def factory(data, r, c):
        if "opt" in shared:                 # constraint bounds (the loads) differ per scenario: new LP skeleton
            shared["opt"].close()
        shared["opt"] = A.HipSubOptimizer(data, r, c)
        return shared["opt"]

    def make_model(sidx):
        pr = acopf.acopf_problem(acopf.scenario_case(base, sidx, 0.95, 1.05), "case3 scenario")
        return A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=100, external_optimizer=factory))

    slps, stats = batch.solve_batch(make_model, 4, rank=0, world=1)
    assert stats["scenarios"] == 4 and stats["converged"] == 4
    objs = [s.problem.obj_val for s in slps]
    assert all(5000 < o < 7000 for o in objs) and len(set(round(o, 3) for o in objs)) == 4
    # same scenarios through the oracle
    from oracle import slp as O
    for sidx, sh in zip(range(4), slps):
        pr = acopf.acopf_problem(acopf.scenario_case(base, sidx, 0.95, 1.05), "case3 scenario")
        mo, so = _oracle_run(pr, algorithm="Line Search", max_iter=100)
        assert abs(mo.obj_val - sh.problem.obj_val) < 1e-6 and so.lp_solves == sh.lp_solves
    shared["opt"].close()

    # stream pool: three scenarios in flight, one handle / HIP stream each, from host threads - same results, same order
    def make_model_pool(sidx):
        pr = acopf.acopf_problem(acopf.scenario_case(base, sidx, 0.95, 1.05), "case3 scenario")
        return A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=100))

    def run(model):
        slp = A.optimize(model)
        slp.optimizer.close()
        return slp

    slps2, stats2 = batch.solve_batch(make_model_pool, 4, rank=0, world=1, run=run, concurrency=3)
    assert [s.lp_solves for s in slps2] == [s.lp_solves for s in slps]
    assert all(s2.problem.obj_val == s1.problem.obj_val for s1, s2 in zip(slps, slps2))
    assert np.array_equal(np.concatenate([s.x for s in slps2]), np.concatenate([s.x for s in slps]))
    for k in ("scenarios", "converged", "iterations", "lp_solves", "restoration_solves", "inf_pr", "inf_du"):
        assert stats2[k] == stats[k]


# ---------------------------------------------------------------------------------------------------------------------------
# Regression: full SLP runs at case118 / case300 size.  Round 1 left a run on record (gpurun_out/fail.log) where a
# case118-sized run ended with status -5 because an LP came back unpolished (path 5) / status OTHER: the interior-point
# iteration was declared "jammed" at a primal residual of 1e-8 (rounding level) one iteration before convergence, and the
# polish of the not-quite-converged iterate failed.  Every LP of these runs must now end in an active-set solve.
@pytest.mark.parametrize("case,load,alg,expect", [
    ("case118", 1.0, "Line Search", (0,)),
    ("case118", 1.0, "Trust Region", (0, 6, -1)),
    ("case300", 0.5, "Line Search", (0,)),
    ("case300", 1.0, "Line Search", (0, 6, -1, 2)),
    ("case1354pegase", 0.5, "Line Search", (0,)),      # BASELINE.json configs[3] size, the bench's C4 instance: terminates LOCALLY_SOLVED
])
def test_slp_run_to_termination_every_lp_polished(case, load, alg, expect):
    from activesetmethods_amd import acopf
    pr = acopf.acopf_problem(acopf.synthetic_case(case, 1, load), case)
    mh, sh = _hip_run(pr, algorithm=alg, max_iter=100)
    bad = [(k, r['status'], r['stats']['path'], r['stats']['polished']) for k, r in enumerate(sh.trace)
           if r['status'] not in (1, 2) or r['stats']['polished'] != 1 or r['stats']['path'] == 5]
    assert not bad, bad
    assert mh.status in expect, (mh.status, sh.iter, sh.lp_solves)       # never -5 (LP solver trouble, slp_line_search.jl:127-133)


def test_case300_sized_lp_parity_normal_and_restoration():
    """C5's size (n = 2382, m = 3889): one normal-phase LP (Line-Search radius) and the restoration LP after the INFEASIBLE
    Trust-Region LP at the same point, HIP vs oracle: status, path, working sets, 1e-9 on step and multipliers."""
    from activesetmethods_amd import acopf
    from tests.util import oracle_solve, hip_solve
    pr = acopf.acopf_problem(acopf.synthetic_case("case300", 1, 0.5), "case300")
    x = pr.x0.copy()
    base = dict(n=pr.n, m=pr.m, j_row=pr.j_row, j_col=pr.j_col, dE=pr.eval_jac_g(x, np.zeros(pr.nnz)), df=pr.eval_grad_f(x, np.zeros(pr.n)),
                f=pr.eval_f(x), E=pr.eval_g(x, np.zeros(pr.m)), x_k=x, c_lb=pr.g_L, c_ub=pr.g_U, v_lb=pr.x_L, v_ub=pr.x_U)
    names = {0: 'warm', 1: 'ipm0+ln', 2: 'ipm1+ln', 3: 'ipm2+ln', 4: 'ipm+face', 5: 'ipm-unpolished', 6: 'ipm-infeasible', 7: 'phase1-infeasible',
             8: 'ipm~+ln', 9: 'ipm+ref', 10: 'ipm-conv'}
    # normal phase, Line-Search radius
    sp = dict(base, delta=1000.0)
    qp, o_out = oracle_solve(sp)
    opt, h_out = hip_solve(sp)
    assert o_out[5] == h_out[5] == 1
    assert names[opt.last_stats()['path']] == o_out[6]['stats']['path']
    for a, b in zip(opt.active_set(), o_out[6]['sets']):
        assert np.array_equal(a, b)
    for k in range(4):
        assert rel_err(h_out[k], o_out[k]) < 1e-9, k
    opt.close()
    # Trust-Region radius: INFEASIBLE, then the restoration LP (non-unique optimum)
    sp = dict(base, delta=0.05)
    qp, o_out = oracle_solve(sp)
    opt, h_out = hip_solve(sp)
    assert o_out[5] == h_out[5]
    if o_out[5] == 2:
        qp, o_out = oracle_solve(sp, True, qp)
        opt, h_out = hip_solve(sp, True, opt)
    assert o_out[5] == h_out[5] == 1
    assert names[opt.last_stats()['path']] == o_out[6]['stats']['path']
    for a, b in zip(opt.active_set(), o_out[6]['sets']):
        assert np.array_equal(a, b)
    for k in range(4):
        assert rel_err(h_out[k], o_out[k]) < 1e-9, k
    opt.close()


def test_case300_scenario_batch_all_converge():
    """BASELINE.json configs[4] at reduced count: 8 scenarios of the case300-sized grid (the C5 workload of bench.py: load
    scale 0.5, loads x U(0.9,1.1) per scenario) through the batch path on one GPU, three in flight; every scenario must
    converge (status 0) with every LP polished, and the merged statistics must say so."""
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf, batch
    base = acopf.synthetic_case("case300", 1, 0.5)

    def make_model(sidx):
        pr = acopf.acopf_problem(acopf.scenario_case(base, sidx), "case300-sized scenario %d" % sidx)
        return A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=100))

    def run(model):
        slp = A.optimize(model)
        slp.optimizer.close()
        return slp

    slps, stats = batch.solve_batch(make_model, 8, rank=0, world=1, run=run, concurrency=3)
    assert stats["scenarios"] == 8 and stats["converged"] == 8, stats
    assert all(s.ret == 0 for s in slps)
    assert all(r['status'] in (1, 2) and r['stats']['polished'] == 1 for s in slps for r in s.trace)
    assert stats["inf_pr"] <= 0.01 and stats["inf_du"] <= 0.01


def test_converged_iterate_last_resort_scenario_27():
    """case300-sized scenario 27 (the one of the first 64 scenarios of the C5 batch that used to stop with an unpolished LP): at LP 51
    the interior-point iterate converges to 1e-13 in all three measures, but the vertex is so nearly degenerate that every active-set solve
    misses its own primal test by ~1e-6.  The converged iterate is then the answer ('ipm-conv', path 10: counted as non-canonical), the
    run goes on and terminates with status 0.  The oracle takes the same path on that LP and agrees on the step."""
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf
    from tests.util import oracle_solve, rel_err
    base = acopf.synthetic_case("case300", 1, 0.5)
    case = acopf.scenario_case(base, 27)
    pr = acopf.acopf_problem(case, "case300-sized scenario 27")                       # host callbacks: to rebuild the LP for the oracle
    prd = acopf.function_model(case).to_problem("case300-sized scenario 27")          # the batch's own model (device evaluator)
    slp = A.optimize(A.Model.from_problem(prd, A.Parameters(algorithm="Line Search", max_iter=200, device_eval=True)))
    slp.optimizer.close()
    assert slp.ret == 0
    assert all(r['status'] == 1 for r in slp.trace)
    conv = [k for k, r in enumerate(slp.trace) if r['stats']['path'] == 10]
    # (with the interior-point start of round 4 the LP that needed the last resort is polished like the others: the path itself is pinned by
    # test_last_resort_answer_equals_the_oracle below and, on restoration LPs, by test_restoration_run_at_load_0_8_has_no_unpolished_lp)
    assert len(conv) <= 2 and all(r['stats']['path'] in (0, 1, 2, 3, 4, 10) for r in slp.trace), [r['stats']['path'] for r in slp.trace]
    for k in conv[:1]:
        rec = slp.trace[k]
        x = rec['x']
        sp = dict(n=pr.n, m=pr.m, j_row=pr.j_row, j_col=pr.j_col, dE=pr.eval_jac_g(x, np.zeros(pr.nnz)), df=pr.eval_grad_f(x, np.zeros(pr.n)),
                  f=pr.eval_f(x), E=pr.eval_g(x, np.zeros(pr.m)), x_k=x.copy(), c_lb=pr.g_L, c_ub=pr.g_U, v_lb=pr.x_L, v_ub=pr.x_U, delta=rec['delta'])
        qp, o_out = oracle_solve(sp)
        assert o_out[5] == 1 and o_out[6]['stats']['path'] == 'ipm-conv', o_out[6]['stats']
        assert abs(sp['df'] @ (o_out[0] - rec['p'])) <= 1e-8 * max(1.0, abs(sp['df'] @ rec['p']))
        assert rel_err(rec['p'], o_out[0]) < 1e-6
        # the non-canonical answer against an independent LP code: HiGHS optimal value of the same LP, and the GPU's step is feasible for it
        from oracle import sparse_lp
        lp = sparse_lp.build(pr.n, pr.m, pr.j_row, pr.j_col, sp['dE'], sp['df'], sp['E'], pr.g_L, pr.g_U, pr.x_L, pr.x_U, x, rec['delta'], False)
        st, obj, _, _, _ = sparse_lp.solve_highs(lp)
        # (1e-6: the last resort accepts an iterate converged to 1e-8 in the scaled measures since round 4 - measured 2.6e-7 here - INTEGRATION.md section 1)
        assert st == 1 and abs(sp['df'] @ rec['p'] - obj) <= 1e-6 * max(1.0, abs(obj))
        # (feasibility in the caller's units: this weakest path hands out an interior iterate converged to 1e-10 in the scaled measures -
        # measured 5.1e-7 absolute on the equality rows, whose entries are of order 10; an active-set answer sits at 1e-13)
        Ap = lp['A_ub'] @ rec['p'] - lp['b_ub']
        assert Ap.max(initial=0.0) <= 5e-6 and np.abs(lp['A_eq'] @ rec['p'] - lp['b_eq']).max(initial=0.0) <= 5e-6      # (measured 1.02e-6 in round 4, 5.1e-7 in round 3)
        assert np.all(rec['p'] >= lp['bounds'][:pr.n, 0] - 1e-9) and np.all(rec['p'] <= lp['bounds'][:pr.n, 1] + 1e-9)


@pytest.mark.parametrize("seed,n,m", [(5, 60, 40), (9, 120, 90)])
def test_last_resort_answer_equals_the_oracle(seed, n, m, monkeypatch):
    """The last resort of the LP solve (asm_solve_stats.path 10, 'ipm-conv'): with every active-set attempt made to fail - asm_test_no_polish
    in the library, eqp_loop / face_polish patched in the oracle - both end on the converged interior iterate: status OPTIMAL, the same
    iteration count, the same point to 1e-8 and, against the polished answer of the same LP, to 1e-6."""
    from oracle import lp_solver as L
    from tests.util import random_subproblem, oracle_solve, hip_solve, rel_err
    sp = random_subproblem(seed, n, m, 0.3, 0.0, 2)
    qp, ref = oracle_solve(sp)
    assert ref[5] == 1
    monkeypatch.setattr(L, "eqp_loop", lambda *a, **k: (False, None, None, None, None))
    monkeypatch.setattr(L, "face_polish", lambda *a, **k: (None, None, None, None, None))
    qp2, o_out = oracle_solve(sp)
    assert o_out[5] == 1 and o_out[6]['stats']['path'] == 'ipm-conv'
    from activesetmethods_amd.subproblem import QpData, HipSubOptimizer
    opt = HipSubOptimizer(QpData(sp['df'], sp['f'], sp['dE'], sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub']), sp['j_row'], sp['j_col'])
    assert opt._lib.asm_test_no_polish(opt._h, 1) == 0
    h_out = opt.sub_optimize(sp['x_k'], sp['delta'], False)
    st = opt.last_stats()
    assert h_out[5] == 1 and st['path'] == 10 and st['polished'] == 1
    assert st['ipm_iters'] == o_out[6]['stats']['ipm_iters']
    assert rel_err(h_out[0], o_out[0]) < 1e-8 and rel_err(h_out[1], o_out[1]) < 1e-6
    assert rel_err(h_out[0], ref[0]) < 1e-6
    assert opt._lib.asm_test_no_polish(opt._h, 0) == 0
    h2 = opt.sub_optimize(sp['x_k'], sp['delta'], False)                 # the hook is off again: the polished answer
    assert opt.last_stats()['path'] in (1, 2, 3) and rel_err(h2[0], ref[0]) < 1e-10
    opt.close()


def test_scenario_batch_is_independent_of_the_stream_pool():
    """Eight case300-sized scenarios solved one after the other and three at a time (each in-flight scenario on its own handle = HIP stream):
    the same iterates, bit for bit - x, multipliers, iteration and LP counts of every scenario.  (The reductions of the kernels are
    fixed-order; nothing of one scenario's solve depends on what else the GPU is doing.)"""
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf, batch
    base = acopf.synthetic_case("case300", 1, 0.5)

    def make_model(sidx):
        pr = acopf.function_model(acopf.scenario_case(base, sidx)).to_problem("case300-sized scenario %d" % sidx)
        return A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=100, device_eval=True))

    def run(model):
        slp = A.optimize(model)
        slp.optimizer.close()
        return slp, model

    seq, _ = batch.solve_batch(make_model, 8, rank=0, world=1, run=lambda m: run(m)[0], concurrency=1)
    par, _ = batch.solve_batch(make_model, 8, rank=0, world=1, run=lambda m: run(m)[0], concurrency=3)
    for a, b in zip(seq, par):
        assert a.ret == b.ret == 0 and a.iter == b.iter and a.lp_solves == b.lp_solves
        assert np.array_equal(a.x, b.x) and np.array_equal(a.lam, b.lam)
        assert [r['stats']['path'] for r in a.trace] == [r['stats']['path'] for r in b.trace]


def test_device_side_step_of_null_space_iterations_is_bit_identical():
    """The null-space iterations apply their step on the device and are checked with the next measures (one read-back per iteration,
    k_ns_update_dev); ASM_NS_DEFER=0 is the host-side step of the rounds before.  Same arithmetic: the two runs must agree bit for bit
    (knobs are read once per process, hence two child processes, one after the other)."""
    import hashlib, os, subprocess, sys
    code = ("import hashlib, numpy as np, activesetmethods_amd as A\n"
            "from activesetmethods_amd import acopf\n"
            "pr = acopf.function_model(acopf.synthetic_case('case118', 1, 0.5)).to_problem('case118-sized')\n"
            "m = A.Model.from_problem(pr, A.Parameters(algorithm='Line Search', max_iter=6, device_eval=True))\n"
            "s = A.optimize(m)\n"
            "hh = hashlib.sha256()\n"
            "for r in s.trace:\n"
            "    hh.update(np.ascontiguousarray(r['p'], dtype=np.float64).tobytes()); hh.update(np.ascontiguousarray(r['lam'], dtype=np.float64).tobytes())\n"
            "print('HASH', hh.hexdigest(), len(s.trace), sum(r.get('ns_iters', 0) for r in s.trace))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(**knobs):
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), **knobs)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("HASH")][-1].split()

    dev, host = run(ASM_NS_DEFER="1"), run(ASM_NS_DEFER="0")
    assert dev[1] == host[1], (dev, host)
    assert int(dev[2]) >= 5
    # the fall-back: with an accuracy bound nothing can meet (ASM_NS_RERR) every null-space iteration is rejected - the device leaves the iterate
    # alone, the host redoes the iteration in row form - again the same numbers either way, and (the row form converging to the same LP
    # answers) the same SLP trace to the parity tolerance
    dev_f, host_f = run(ASM_NS_DEFER="1", ASM_NS_RERR="1e-300"), run(ASM_NS_DEFER="0", ASM_NS_RERR="1e-300")
    assert dev_f[1] == host_f[1], (dev_f, host_f)
    assert dev_f[2] == dev[2]
