"""Lockstep scenario batch against the per-scenario path (GPU): bit-identity on a few case300-sized scenarios, then throughput."""
import sys, time, os
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import activesetmethods_amd as A
from activesetmethods_amd import acopf, batch, _lib

case = sys.argv[1] if len(sys.argv) > 1 else "case300"
nsc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
slots = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["8"])]
check = int(sys.argv[4]) if len(sys.argv) > 4 else 1
base = acopf.synthetic_case(case, 1, 0.5)
t0 = time.perf_counter()
prs = [acopf.function_model(acopf.scenario_case(base, s)).to_problem("s%d" % s) for s in range(nsc)]
print("built %d scenario models in %.1f s" % (nsc, time.perf_counter() - t0), flush=True)
par = A.Parameters(algorithm="Line Search", max_iter=100, device_eval=True)

for B in slots:
    hb = batch.HipBatch(prs[0], min(B, nsc), 0)
    t0 = time.perf_counter()
    runs, st, bst = batch.solve_batch_lockstep(prs, par, B, batch=hb)
    dt = time.perf_counter() - t0
    print("slots %d: %.2f s -> %.2f solves/s; converged %d/%d; iters %s" % (B, dt, nsc / dt, sum(r.ret == 0 for r in runs), nsc, [r.iter for r in runs][:16]), flush=True)
    print("   batch stats", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in bst.items()}, "merge ratio %.2f" % (bst["ops"] / max(bst["launches"], 1)), flush=True)
    J = hb.ns_basis()
    hb.close()

if check:
    # per-scenario path: one handle, the same reference basis columns, asm_slp_run per scenario
    lib = _lib.load()
    opt = A.HipSubOptimizer(A.QpData(np.zeros(prs[0].n), 0.0, np.zeros(prs[0].nnz), np.zeros(prs[0].m), prs[0].g_L, prs[0].g_U, prs[0].x_L, prs[0].x_U), prs[0].j_row, prs[0].j_col)
    opt.eval_setup(prs[0].function_model)
    sp = batch.slp_params(par)
    worst = 0.0
    t0 = time.perf_counter()
    for s, pr in enumerate(prs[:check if check > 1 else nsc]):
        opt.set_bounds(A.QpData(None, 0.0, None, None, pr.g_L, pr.g_U, pr.x_L, pr.x_U))
        Jc = np.ascontiguousarray(J, np.int32)
        assert lib.asm_sublp_set_ns_basis(opt._h, _lib.i32ptr(Jc), len(Jc)) == 0
        x = np.empty(pr.n); lam = np.empty(max(pr.m, 1)); mU = np.empty(pr.n); mL = np.empty(pr.n); g = np.empty(max(pr.m, 1))
        res = _lib.SlpResult()
        x0 = np.ascontiguousarray(pr.x0, np.float64)
        rc = lib.asm_slp_run(opt._h, C.byref(sp), _lib.dptr(x0), _lib.dptr(x), _lib.dptr(lam), _lib.dptr(mU), _lib.dptr(mL), _lib.dptr(g), C.byref(res))
        assert rc == 0, lib.asm_last_error(opt._h)
        r = runs[s]
        same = (np.array_equal(x, r.x) and np.array_equal(lam[:pr.m], r.lam) and res.iter == r.iter and res.status == r.ret)
        worst = max(worst, float(np.max(np.abs(x - r.x))))
        print("scenario %d: per-scenario iter %d status %d | batch iter %d status %d | bit-identical %s" % (s, res.iter, res.status, r.iter, r.ret, same), flush=True)
    print("per-scenario path: %.2f s; worst |x - x_batch| = %.3e" % (time.perf_counter() - t0, worst))
