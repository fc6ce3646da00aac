"""Lockstep scenario batch (include/asm_hip.h: asm_batch_*, asm_slp_run) against the per-handle path, through the C ABI.

The reference has no batching (one Optimizer <-> one Model <-> one SLP object, src/MOI_wrapper.jl:1093-1152): what is pinned here is that
the batch is a pure re-scheduling of the per-scenario solves - the same kernels with the same arguments, merged across scenarios - i.e.
every output equals the per-handle call's BIT FOR BIT, whatever the number of slots and whatever paths the scenarios take."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _native_run(opt, pr, par, J=None):
    from activesetmethods_amd import _lib, batch
    lib = _lib.load()
    if J is not None:
        Jc = np.ascontiguousarray(J, np.int32)
        assert lib.asm_sublp_set_ns_basis(opt._h, _lib.i32ptr(Jc), len(Jc)) == 0
    sp = batch.slp_params(par)
    x = np.empty(pr.n); lam = np.empty(max(pr.m, 1)); mU = np.empty(pr.n); mL = np.empty(pr.n); g = np.empty(max(pr.m, 1))
    res = _lib.SlpResult()
    x0 = np.ascontiguousarray(pr.x0, np.float64)
    rc = lib.asm_slp_run(opt._h, C.byref(sp), _lib.dptr(x0), _lib.dptr(x), _lib.dptr(lam), _lib.dptr(mU), _lib.dptr(mL), _lib.dptr(g), C.byref(res))
    assert rc == 0, lib.asm_last_error(opt._h)
    return batch.NativeRun(res, x, lam[:pr.m], mU, mL, g[:pr.m])


def _handle_for(pr):
    import activesetmethods_amd as A
    opt = A.HipSubOptimizer(A.QpData(np.zeros(pr.n), 0.0, np.zeros(pr.nnz), np.zeros(pr.m), pr.g_L, pr.g_U, pr.x_L, pr.x_U), pr.j_row, pr.j_col)
    opt.eval_setup(pr.function_model)
    return opt


def test_native_slp_driver_equals_the_python_driver():
    """asm_slp_run restates run!(::SlpLS) (slp_line_search.jl:78-215) inside the library; the Python SlpLS of activesetmethods_amd/slp.py makes
    the same library calls statement by statement: same iterates bit for bit, same counts, same status."""
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf
    pr = acopf.function_model(acopf.synthetic_case("case118", 1, 1.0)).to_problem("case118-sized")
    par = A.Parameters(algorithm="Line Search", max_iter=60, device_eval=True)
    slp = A.optimize(A.Model.from_problem(pr, par))
    slp.optimizer.close()
    opt = _handle_for(pr)
    run = _native_run(opt, pr, par)
    opt.close()
    assert run.ret == slp.ret and run.iter == slp.iter and run.lp_solves == slp.lp_solves
    assert run.restoration_solves == sum(1 for r in slp.trace if r["fr"])
    assert np.array_equal(run.x, slp.x) and np.array_equal(run.lam, slp.lam)
    assert np.array_equal(run.mult_x_U, slp.mult_x_U) and np.array_equal(run.mult_x_L, slp.mult_x_L)
    assert run.obj_val == slp.problem.obj_val
    hist = [0] * 12
    for r in slp.trace:
        hist[r["stats"]["path"]] += 1
    assert run.paths == hist


def _perturbed(sp, seed):
    """The same pattern and row kinds with other values: another instance for the same LP skeleton."""
    rng = np.random.default_rng(1000 + seed)
    q = dict(sp)
    q["dE"] = sp["dE"] * (1.0 + 0.05 * rng.standard_normal(len(sp["dE"])))
    q["df"] = rng.standard_normal(sp["n"])
    q["E"] = sp["E"] + 0.01 * rng.standard_normal(sp["m"])
    q["x_k"] = sp["x_k"] + 0.02 * rng.standard_normal(sp["n"])
    shift = 0.01 * rng.standard_normal(sp["m"])
    q["c_lb"] = sp["c_lb"] + shift
    q["c_ub"] = sp["c_ub"] + shift
    return q


@pytest.mark.parametrize("kind", ["equality_rich", "dense"])
def test_batch_sublp_solve_equals_handle_solves_bit_for_bit(kind):
    """asm_batch_sublp_solve = asm_sublp_solve with a leading scenario dimension: five instances of one skeleton in lockstep, normal and
    restoration phase mixed in one batch (the scenarios take different paths through the solver), and a second round on the same
    slots (retained state per slot) - every output equals the per-handle call's bit for bit."""
    from activesetmethods_amd import batch, problems
    from activesetmethods_amd.subproblem import QpData, HipSubOptimizer
    from tests.util import equality_rich_subproblem, random_subproblem
    base = equality_rich_subproblem(3) if kind == "equality_rich" else random_subproblem(5, 60, 40, density=1.0, n_range=3)
    sps = [_perturbed(base, s) for s in range(5)]
    fr = np.array([0, 0, 1, 0, 1], np.int32)

    class _P:            # what HipBatch needs from a problem: pattern, bounds, a function model (the sub-LP entry does not evaluate)
        pass
    pr = _P()
    pr.n, pr.m, pr.j_row, pr.j_col = base["n"], base["m"], base["j_row"], base["j_col"]
    pr.g_L, pr.g_U, pr.x_L, pr.x_U = base["c_lb"], base["c_ub"], base["v_lb"], base["v_ub"]
    fm = problems.synthetic_dense_function_model(8, 4)      # any model: only its flattened store is uploaded
    hb = batch.HipBatch.__new__(batch.HipBatch)
    # batch without an evaluator: create + setup only
    from activesetmethods_amd import _lib
    lib = _lib.load()
    hb._lib, hb._C, hb._err = lib, C, RuntimeError
    hb.n, hb.m, hb.n_slots = pr.n, pr.m, 5
    hb._b = C.c_void_p()
    assert lib.asm_batch_create(0, 5, C.byref(hb._b)) == 0
    f64 = lambda a: np.ascontiguousarray(a, np.float64)
    jr, jc = np.ascontiguousarray(pr.j_row, np.int64), np.ascontiguousarray(pr.j_col, np.int64)
    gl, gu, xl, xu = map(f64, (pr.g_L, pr.g_U, pr.x_L, pr.x_U))
    assert lib.asm_batch_setup(hb._b, pr.n, pr.m, len(jr), _lib.i64ptr(jr), _lib.i64ptr(jc), _lib.dptr(gl), _lib.dptr(gu), _lib.dptr(xl), _lib.dptr(xu)) == 0
    opts = []
    for rnd in range(2):
        if rnd == 1:
            sps = [_perturbed(base, 10 + s) for s in range(5)]
        stack = lambda k: np.stack([sp[k] for sp in sps])
        out = hb.sublp_solve(stack("dE"), stack("df"), np.array([sp["f"] for sp in sps]), stack("E"), stack("x_k"), np.array([sp["delta"] for sp in sps]), fr,
                             bounds=(stack("c_lb"), stack("c_ub"), stack("v_lb"), stack("v_ub")))
        for s, sp in enumerate(sps):
            data = QpData(sp["df"], sp["f"], sp["dE"], sp["E"], sp["c_lb"], sp["c_ub"], sp["v_lb"], sp["v_ub"])
            if rnd == 0:
                opts.append(HipSubOptimizer(QpData(base["df"], 0.0, base["dE"], base["E"], base["c_lb"], base["c_ub"], base["v_lb"], base["v_ub"]), sp["j_row"], sp["j_col"]))
            opts[s].set_bounds(data)
            ref = opts[s].sub_optimize(sp["x_k"], sp["delta"], bool(fr[s]))
            assert ref[5] == out[5][s]
            for a, b in zip(ref[:4], (out[0][s], out[1][s], out[2][s], out[3][s])):
                assert np.array_equal(a, b)
            assert np.array_equal(ref[4].raw[:2 * pr.m], out[4][s], equal_nan=True)
            assert opts[s].last_stats()["path"] == hb.slot_stats(s)["path"]
    st = hb.stats()
    assert st["ops"] > st["launches"]            # launches were merged across the slots
    for o in opts:
        o.close()
    hb.close()


def test_batched_slp_runs_equal_per_scenario_runs_bit_for_bit():
    """Eight case300-sized scenarios (the C5 workload of bench.py) through asm_batch_slp_run with eight slots in two groups, with three slots (slots take
    the next scenario when they finish one) and one by one through asm_slp_run on a single handle: identical iterates, multipliers,
    iteration / LP counts and solver paths, all converged; with eight slots most launches are shared."""
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf, batch
    base = acopf.synthetic_case("case300", 1, 0.5)
    prs = [acopf.function_model(acopf.scenario_case(base, s)).to_problem("case300-sized scenario %d" % s) for s in range(8)]
    par = A.Parameters(algorithm="Line Search", max_iter=100, device_eval=True)
    hb = batch.HipBatch(prs[0], 8, groups=2)             # two groups of four slots: two streams, two host threads
    assert hb.groups == 2
    runs8, stats, bst = batch.solve_batch_lockstep(prs, par, 8, batch=hb)
    J = hb.ns_basis()
    hb.close()
    assert stats["scenarios"] == 8 and stats["converged"] == 8, stats
    assert len(J) > 0 and bst["ops"] >= 2 * bst["launches"], bst
    hb3 = batch.HipBatch(prs[0], 3)
    hb3.set_ns_basis(J)
    runs3, _, _ = batch.solve_batch_lockstep(prs, par, 3, batch=hb3)
    hb3.close()
    opt = _handle_for(prs[0])
    for s, pr in enumerate(prs):
        opt.set_bounds(A.QpData(None, 0.0, None, None, pr.g_L, pr.g_U, pr.x_L, pr.x_U))
        one = _native_run(opt, pr, par, J)
        for r in (runs8[s], runs3[s]):
            assert r.ret == one.ret == 0 and r.iter == one.iter and r.lp_solves == one.lp_solves and r.paths == one.paths
            assert np.array_equal(r.x, one.x) and np.array_equal(r.lam, one.lam)
            assert np.array_equal(r.mult_x_U, one.mult_x_U) and np.array_equal(r.mult_x_L, one.mult_x_L)
            assert r.obj_val == one.obj_val
    opt.close()
