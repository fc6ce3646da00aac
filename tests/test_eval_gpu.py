"""Device-side evaluators and SLP reductions against the host restatement of the MOI
wrapper's evaluator block (activesetmethods_amd/moi_evaluator.py, src/MOI_wrapper.jl:683-944) and the oracle's norms / merit
functions (src/algorithms/common.jl:35-98, src/algorithms/slp.jl:79-147), through the C ABI."""
import numpy as np
import pytest

from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _random_function_model(seed, n=30, sense="MIN_SENSE"):
    from tests.util import random_function_model
    return random_function_model(seed, n, sense)


def _optimizer_for(pr):
    from activesetmethods_amd import QpData, HipSubOptimizer
    z = np.zeros
    return HipSubOptimizer(QpData(z(pr.n), 0.0, z(pr.nnz), z(pr.m), pr.g_L, pr.g_U, pr.x_L, pr.x_U), pr.j_row, pr.j_col)


@pytest.mark.parametrize("seed,sense", [(1, "MIN_SENSE"), (2, "MAX_SENSE"), (3, "FEASIBILITY_SENSE"), (4, "MIN_SENSE")])
def test_affine_quadratic_evaluator_is_bit_identical(seed, sense):
    """f, grad f, g and the Jacobian values (duplicates and all, in j_str order) from the device kernels equal the host
    evaluator bit for bit: same term order, no fused multiply-add."""
    fm = _random_function_model(seed, sense=sense)
    pr = fm.to_problem()
    opt = _optimizer_for(pr)
    opt.eval_setup(fm)
    rng = np.random.default_rng(seed + 50)
    for _ in range(3):
        x = rng.uniform(-1.0, 1.0, pr.n)
        f, df, E = opt.eval_functions(x)
        assert f == pr.eval_f(x)
        assert np.array_equal(df, pr.eval_grad_f(x, np.zeros(pr.n)))
        assert np.array_equal(E, pr.eval_g(x, np.zeros(pr.m)))
        assert np.array_equal(opt.jacobian_values(), pr.eval_jac_g(x, np.zeros(pr.nnz)))
        ft, Et = opt.eval_constraints(0.5 * x)
        assert ft == pr.eval_f(0.5 * x) and np.array_equal(Et, pr.eval_g(0.5 * x, np.zeros(pr.m)))
        # ... and the oracle's independent restatement of MOI_wrapper.jl:776-944 (oracle/moi_eval.py: no code shared with the product)
        from tests.util import oracle_wrapper_model, oracle_evaluate
        fo, go, Eo, dEo, j_str = oracle_evaluate(oracle_wrapper_model(fm), x)
        assert f == fo and np.array_equal(df, go) and np.array_equal(E, Eo) and np.array_equal(opt.jacobian_values(), dEo)
        assert [tuple(t) for t in zip(pr.j_row, pr.j_col)] == j_str
    opt.close()


@pytest.mark.parametrize("case,seed", [("case118", 2), ("case1354pegase", 1)])
def test_acopf_device_evaluator(case, seed):
    """Polar ACOPF through the wrapper's lists + the Ohm's-law NLP-block kernel: the affine / quadratic rows bit-identical,
    the trigonometric rows and their Jacobian values within 1e-13 relative (device sin / cos vs NumPy's).  case118 size and the
    size of BASELINE.json configs[3] (n = 11192, m = 18637, 64233 Jacobian entries)."""
    from activesetmethods_amd import acopf
    fm = acopf.function_model(acopf.synthetic_case(case, seed))
    pr = fm.to_problem(case)
    opt = _optimizer_for(pr)
    opt.eval_setup(fm)
    rng = np.random.default_rng(7)
    x = pr.x0 + 0.02 * rng.standard_normal(pr.n)
    f, df, E = opt.eval_functions(x)
    Eh = pr.eval_g(x, np.zeros(pr.m)); dEh = pr.eval_jac_g(x, np.zeros(pr.nnz)); dEd = opt.jacobian_values()
    k = fm.nlp_constraint_offset
    nf = fm.flatten()["nnz_functions"]
    assert f == pr.eval_f(x) and np.array_equal(df, pr.eval_grad_f(x, np.zeros(pr.n)))
    assert np.array_equal(E[:k], Eh[:k]) and np.array_equal(dEd[:nf], dEh[:nf])
    assert rel_err(E[k:], Eh[k:]) < 1e-13 and rel_err(dEd[nf:], dEh[nf:]) < 1e-13
    if case == "case118":            # the same against the oracle's restatement of the wrapper block (pure Python: the small size only)
        from tests.util import oracle_wrapper_model, oracle_evaluate
        fo, go, Eo, dEo, j_str = oracle_evaluate(oracle_wrapper_model(fm), x)
        assert f == fo and np.array_equal(df, go) and np.array_equal(E[:k], Eo[:k]) and np.array_equal(dEd[:nf], dEo[:nf])
        assert [tuple(t) for t in zip(pr.j_row, pr.j_col)] == j_str
    opt.close()


def test_dense_quadratic_device_evaluator():
    from activesetmethods_amd import problems
    fm = problems.synthetic_dense_function_model(96, 40)
    pr = fm.to_problem()
    opt = _optimizer_for(pr)
    opt.eval_setup(fm)
    x = np.random.default_rng(3).uniform(-0.5, 0.5, pr.n)
    f, df, E = opt.eval_functions(x)
    assert f == pr.eval_f(x) and np.array_equal(df, pr.eval_grad_f(x, np.zeros(pr.n)))
    assert rel_err(E, pr.eval_g(x, np.zeros(pr.m))) < 1e-13
    assert rel_err(opt.jacobian_values(), pr.eval_jac_g(x, np.zeros(pr.nnz))) < 1e-14
    opt.close()


def test_slp_reductions_on_device():
    """KT_residuals, norm_violations (Inf and 1), norm_complementarity, compute_phi and compute_derivative (both phases) on the
    device against the oracle's / the host driver's formulas."""
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf
    from oracle import slp as O
    from oracle.subproblem import compute_jacobian_matrix
    fm = acopf.function_model(acopf.synthetic_case("case118", 4))
    pr = fm.to_problem("case118")
    rng = np.random.default_rng(11)
    x = pr.x0 + 0.05 * rng.standard_normal(pr.n)
    opt = _optimizer_for(pr)
    opt.eval_setup(fm)
    f, df, E = opt.eval_functions(x)
    lam = rng.standard_normal(pr.m); mU = 0.1 * rng.standard_normal(pr.n); mL = 0.1 * rng.standard_normal(pr.n)
    J, _ = compute_jacobian_matrix(pr.m, pr.n, pr.j_row - 1, pr.j_col - 1, pr.eval_jac_g(x, np.zeros(pr.nnz)))
    ref = (O.norm_violations(E, pr.g_L, pr.g_U, x, pr.x_L, pr.x_U, np.inf), O.norm_violations(E, pr.g_L, pr.g_U, x, pr.x_L, pr.x_U, 1),
           O.KT_residuals(df, lam, mU, mL, J), O.norm_complementarity(E, pr.g_L, pr.g_U, lam))
    got = opt.slp_norms(lam, mU, mL)
    for a, b in zip(got, ref):
        assert abs(a - b) <= 1e-12 * max(1.0, abs(b)), (got, ref)
    # merit function and its directional derivative: the host driver's formulas (activesetmethods_amd/slp.py) on the same data
    mdl = A.Model.from_problem(pr, A.Parameters())
    host = A.SlpLS(mdl)
    host.x = x.copy(); host.f, host.df, host.E = f, df.copy(), E.copy()
    host.nu = np.abs(rng.standard_normal(pr.m)); host.p = 0.01 * rng.standard_normal(pr.n)
    host.prim_infeas = ref[0]
    both = (pr.g_L > -np.inf) & (pr.g_U < np.inf)
    host.p_slack = {i: ([abs(rng.standard_normal()), abs(rng.standard_normal())] if both[i] else [abs(rng.standard_normal())]) for i in range(pr.m)}
    for fr in (False, True):
        host.feasibility_restoration = fr
        for alpha in (0.0, 1.0, 0.37):
            want = host.compute_phi(host.x, alpha, host.p)
            got = opt.slp_merit(0, alpha, host.p, host.nu, host.p_slack, fr, host.prim_infeas)
            assert abs(got - want) <= 1e-11 * max(1.0, abs(want)), (fr, alpha, got, want)
        want = host.compute_derivative()
        got = opt.slp_merit(1, 0.0, host.p, host.nu, host.p_slack, fr, host.prim_infeas)
        assert abs(got - want) <= 1e-11 * max(1.0, abs(want)), (fr, got, want)
    # ... and directly against the oracle's restatement of slp.jl:79-147 (oracle/slp.py: compute_phi, compute_derivative)
    om = O.Model(pr.n, pr.m, pr.x_L, pr.x_U, pr.g_L, pr.g_U, pr.j_str, pr.eval_f, pr.eval_g, pr.eval_grad_f, pr.eval_jac_g, O.Parameters())
    osl = O.SlpLS(om)
    osl.x = x.copy(); osl.f = f; osl.df = df.copy(); osl.E = E.copy(); osl.nu = host.nu.copy(); osl.p = host.p.copy()
    osl.prim_infeas = ref[0]; osl.p_slack = {i: list(v) for i, v in host.p_slack.items()}
    for fr in (False, True):
        osl.feasibility_restoration = fr
        for alpha in (0.0, 1.0, 0.37):
            want = osl.compute_phi(osl.x, alpha, osl.p)
            got = opt.slp_merit(0, alpha, host.p, host.nu, host.p_slack, fr, host.prim_infeas)
            assert abs(got - want) <= 1e-11 * max(1.0, abs(want)), ("oracle", fr, alpha, got, want)
        want = osl.compute_derivative()
        got = opt.slp_merit(1, 0.0, host.p, host.nu, host.p_slack, fr, host.prim_infeas)
        assert abs(got - want) <= 1e-11 * max(1.0, abs(want)), ("oracle", fr, got, want)
    opt.close()


def _toy_function_model():
    """The toy NLP of test/ext_solver.jl:12-18 with its three nonlinear rows stated as quadratic functions."""
    from activesetmethods_amd.moi_evaluator import FunctionModel, ScalarFunction
    fm = FunctionModel(2)
    fm.objective = ScalarFunction(0.0, [(1.0, 1)], [(2.0, 1, 1)])                      # X^2 + X
    fm.add_constraint(ScalarFunction(0.0, [(1.0, 1)]), "ge", -2.0)                      # X >= -2
    fm.add_constraint(ScalarFunction(0.0, [(-1.0, 1)], [(2.0, 1, 1)]), "eq", 2.0)       # X^2 - X == 2
    fm.add_constraint(ScalarFunction(0.0, [], [(1.0, 1, 2)]), "eq", 1.0)                # X Y == 1
    fm.add_constraint(ScalarFunction(0.0, [], [(1.0, 1, 2)]), "ge", 0.0)                # X Y >= 0
    return fm


@pytest.mark.parametrize("alg", ["Line Search", "Trust Region"])
def test_toy_with_device_evaluation_reaches_the_known_answer(alg):
    """test/runtests.jl:11-13 with every evaluation and reduction on the GPU: X = Y = -1, LOCALLY_SOLVED; same run as with the
    host evaluator."""
    import activesetmethods_amd as A
    runs = []
    for dev in (True, False):
        pr = _toy_function_model().to_problem("toy")
        m = A.Model.from_problem(pr, A.Parameters(algorithm=alg, device_eval=dev))
        s = A.optimize(m)
        runs.append((m, s))
    (md, sd), (mh, sh) = runs
    assert md.status == mh.status == 0
    assert np.allclose(md.x, [-1.0, -1.0], rtol=1e-4)
    assert sd.lp_solves == sh.lp_solves and rel_err(md.x, mh.x) < 1e-9


def test_acopf_slp_run_device_evaluation_matches_host_evaluation():
    """case118-sized ACOPF (FunctionModel + Ohm's-law kernel), Line Search, 15 iterations: the run with device-side
    evaluation, norms and merit reductions follows the run with the host evaluator (same LP statuses and phases, steps within
    1e-8: the two differ only by the last bits of sin / cos and of the reductions)."""
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf
    case = acopf.synthetic_case("case118", 1)
    runs = []
    for dev in (True, False):
        pr = acopf.function_model(case).to_problem("case118")
        m = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=15, device_eval=dev))
        runs.append((m, A.optimize(m)))
    (md, sd), (mh, sh) = runs
    assert len(sd.trace) == len(sh.trace)
    for a, b in zip(sd.trace, sh.trace):
        assert a["status"] == b["status"] and a["fr"] == b["fr"]
        assert rel_err(a["p"], b["p"]) < 1e-8
    assert rel_err(md.x, mh.x) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("fr", [False, True])
def test_device_line_search_equals_trial_by_trial_merit(fr):
    """asm_slp_line_search (trial points evaluated eight per read-back) against the loop of slp_line_search.jl:222-244 built from
    asm_slp_merit calls: same alpha, same merit value, same number of trials - for steps that are accepted at once, after a few
    reductions, after more than one batch of eight, and never (alpha < min_alpha)."""
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf
    case = acopf.synthetic_case("case118", 1, 1.0)
    pr = acopf.function_model(case).to_problem("case118 ls")
    mdl = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=50, device_eval=True))
    slp = A.SlpLS(mdl)
    slp.run(max_lp_solves=3)
    slp.eval_functions()
    opt = slp.optimizer
    rng = np.random.default_rng(11)
    nu = np.abs(rng.standard_normal(pr.m)) + 0.1
    ps = np.abs(rng.standard_normal(2 * pr.m))
    both = (pr.g_L > -np.inf) & (pr.g_U < np.inf)
    ps[1::2][~both] = np.nan
    prim = 0.37
    eta, tau, min_alpha = 0.4, 0.9, 1e-6
    for scale, dd in ((1e-3, -1.0), (0.3, -5.0), (3.0, -50.0), (3.0, -1e9), (0.5, 1e3)):
        p = scale * rng.standard_normal(pr.n)
        phi0 = opt.slp_merit(0, 0.0, p, nu, type("S", (), {"raw": ps})(), fr, prim)
        alpha, trials, ok, phi_a = 1.0, 0, None, None
        while True:
            phi_a = opt.slp_merit(0, alpha, p, nu, type("S", (), {"raw": ps})(), fr, prim)
            trials += 1
            if not (phi_a > phi0 + eta * alpha * dd):
                ok = True
                break
            if alpha < min_alpha:
                ok = False
                break
            alpha *= tau
        got = opt.slp_line_search(p, nu, type("S", (), {"raw": ps})(), fr, prim, phi0, dd, eta, tau, min_alpha)
        assert got[3] == ok and got[0] == alpha and got[2] == trials and got[1] == phi_a, (scale, dd, got, alpha, phi_a, trials, ok)
