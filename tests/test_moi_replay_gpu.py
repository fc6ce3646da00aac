"""The MOI-level glue for the UNMODIFIED reference (`"external_optimizer" => AsmHip.Optimizer`) without a
Julia runtime: activesetmethods_amd/moi_optimizer.py restates the Julia module method by method, oracle/moi_replay.py issues the
reference's exact MOI call sequence (create_model! + both phases of sub_optimize!, subproblem.jl:51-215, 229-542, including the
GreaterThan <-> EqualTo transforms of the slack bounds and the stale-entry rule of range rows).

  * CPU: the LP the optimizer has recorded when `MOI.optimize!` arrives equals the LP of the oracle's restatement (build_lp);
  * GPU: the 6-tuple the reference would get back equals what `asm_sublp_solve` returns for the same sub-problem, call after call."""
import numpy as np
import pytest

from tests.util import random_subproblem, rel_err


def _oracle_data(sp):
    from oracle.subproblem import QpData, compute_jacobian_matrix
    A, stored = compute_jacobian_matrix(sp['m'], sp['n'], sp['j_row'] - 1, sp['j_col'] - 1, sp['dE'])
    return QpData(sp['df'], sp['f'], A, sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub'], stored)


def _calls(seed):
    """A sequence of sub-problems on one pattern: normal, restoration, perturbed normal (entries vanish: stale coefficients of range
    rows must persist), restoration again - every slack-bound path of sub_optimize! (set / transform, both directions)."""
    sp = random_subproblem(seed, 40, 30, 0.3, 0.3, 5)
    rng = np.random.default_rng(seed)
    out = [(sp, False), (sp, True)]
    sp2 = dict(sp); dE = sp['dE'].copy(); dE[rng.random(len(dE)) < 0.3] = 0.0
    sp2['dE'] = dE; sp2['E'] = sp['E'] + 0.05 * rng.standard_normal(sp['m'])
    out += [(sp2, False), (sp2, True)]
    sp3 = dict(sp); sp3['dE'] = rng.standard_normal(len(sp['dE'])) * 0.2; sp3['x_k'] = np.clip(sp['x_k'] + 0.1, -0.9, 0.9)
    out += [(sp3, True), (sp3, False)]
    return out


@pytest.mark.parametrize("seed", [91, 92, 93])
def test_recorded_lp_equals_the_reference_formulation(seed):
    """No GPU: `optimize` is intercepted; what the optimizer recorded from the MOI calls must be the LP of oracle.subproblem.build_lp
    (objective, row types and right-hand sides incl. the extra rows of range constraints, the stale coefficients, column box, slack
    weights / lower bounds / fixed state)."""
    from activesetmethods_amd import moi_optimizer as MOI
    from oracle.moi_replay import QpModelReplay
    from oracle.subproblem import QpModel

    class Recorder(MOI.Optimizer):
        def optimize(self):
            self.status = MOI.OTHER_ERROR

    calls = _calls(seed)
    sp0 = calls[0][0]
    rec = Recorder()
    qp = QpModelReplay(rec, _oracle_data(sp0), sp0['j_row'], sp0['j_col'])
    qp.create_model(sp0['x_k'], sp0['delta'])
    oq = QpModel(_oracle_data(sp0), sp0['j_row'], sp0['j_col'])
    for sp, fr in calls:
        qp.data = _oracle_data(sp); oq.data = _oracle_data(sp)
        qp.sub_optimize(sp['x_k'], sp['delta'], fr)
        lp = oq.build_lp(sp['x_k'], sp['delta'], fr)
        n, R = sp['n'], len(rec.rtype)
        assert R == lp.M and list(rec.rtype) == list(lp.rtype)
        assert np.array_equal(np.array(rec.rhs), lp.r)
        assert np.array_equal(rec.lb, lp.lb) and np.array_equal(rec.ub, lp.ub)
        A = np.zeros((R, n))
        for (r, c), v in rec.coef.items():
            A[r - 1, c - 1] = v
        assert np.array_equal(A, lp.A)
        if fr:
            assert np.array_equal(rec.q, np.zeros(n)) and not any(rec.sfixed)
            # slack columns: MOI creation order -> (row, coefficient) pairs must be the oracle's layout up to the order of columns
            pairs = sorted((r, c, rec.w[k - 1], rec.slo[k - 1]) for r in range(R) for k, c in rec.row_slacks[r])
            want = sorted((int(lp.srow[k]), float(lp.scoef[k]), float(lp.w[k]), float(lp.slo[k])) for k in range(lp.ns))
            assert pairs == want
        else:
            assert np.array_equal(rec.q, lp.q) and all(rec.sfixed) and all(v == 0.0 for v in rec.slo) and all(v == 0.0 for v in rec.w)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [91, 92, 94])
def test_moi_optimizer_returns_what_asm_sublp_solve_returns(seed):
    """The reference's call sequence through the twin of AsmHip.Optimizer (-> asm_sublp_setup + asm_lp_solve) against the
    AbstractSubOptimizer seam (asm_sublp_solve) on the same sub-problems, in order, on one handle each."""
    from activesetmethods_amd import moi_optimizer as MOI
    from activesetmethods_amd.subproblem import QpData, HipSubOptimizer
    from oracle.moi_replay import QpModelReplay
    calls = _calls(seed)
    sp0 = calls[0][0]
    twin = MOI.Optimizer()
    qp = QpModelReplay(twin, _oracle_data(sp0), sp0['j_row'], sp0['j_col'])
    qp.create_model(sp0['x_k'], sp0['delta'])
    opt = HipSubOptimizer(QpData(sp0['df'], sp0['f'], sp0['dE'], sp0['E'], sp0['c_lb'], sp0['c_ub'], sp0['v_lb'], sp0['v_ub']), sp0['j_row'], sp0['j_col'])
    codes = {MOI.OPTIMAL: 1, MOI.INFEASIBLE: 2, MOI.DUAL_INFEASIBLE: 3, MOI.OTHER_ERROR: 4}
    seen = set()
    for sp, fr in calls:
        qp.data = _oracle_data(sp)
        X, lam, mU, mL, ps, st = qp.sub_optimize(sp['x_k'], sp['delta'], fr)
        opt.data = QpData(sp['df'], sp['f'], sp['dE'], sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub'])
        Xh, lh, uh, Lh, psh, sth = opt.sub_optimize(sp['x_k'], sp['delta'], fr)
        assert codes[st] == sth, (fr, st, sth)
        seen.add(sth)
        if sth != 1:
            assert not X.any() and not lam.any()                 # subproblem.jl:532-536
            continue
        assert rel_err(X, Xh) < 1e-9 and rel_err(lam, lh) < 1e-9
        assert rel_err(mU, uh) < 1e-9 and rel_err(mL, Lh) < 1e-9
        if fr:
            for i in range(sp['m']):
                assert rel_err(ps[i], psh[i]) < 1e-9
    assert 1 in seen
    twin.close(); opt.close()
