"""HIP sub-problem solve vs the CPU oracle on the same seeded inputs, through the C ABI
(asm_sublp_setup / asm_sublp_solve).  Bars: status and active sets identical; step, multipliers and
slack values within 1e-10 relative (BASELINE.json north_star)."""
import numpy as np
import pytest

from tests.util import random_subproblem, oracle_solve, hip_solve, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _compare(o_out, h_out, opt, qp_info):
    Xo, lo, uo, Lo, pso, sto, info = o_out
    Xh, lh, uh, Lh, psh, sth = h_out
    assert sth == sto
    if sto != 1:
        assert not Xh.any() and not lh.any()
        return
    rows, bnd, sl = opt.active_set()
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
