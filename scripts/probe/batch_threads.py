"""Two lockstep batches on two host threads (each its own stream) against one batch of all slots: does host work of one overlap the device work of the other?"""
import sys, time, os
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import activesetmethods_amd as A
from activesetmethods_amd import acopf, batch

nsc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 2
slots = int(sys.argv[3]) if len(sys.argv) > 3 else nsc // groups
base = acopf.synthetic_case("case300", 1, 0.5)
prs = [acopf.function_model(acopf.scenario_case(base, s)).to_problem("s%d" % s) for s in range(nsc)]
par = A.Parameters(algorithm="Line Search", max_iter=100, device_eval=True)
hbs = [batch.HipBatch(prs[0], slots, 0) for _ in range(groups)]
wp = prs[:slots]
for hb in hbs:      # warm-up: allocations + reference basis
    hb.slp_run(np.stack([p.g_L for p in wp]), np.stack([p.g_U for p in wp]), np.stack([p.x_L for p in wp]), np.stack([p.x_U for p in wp]), np.stack([p.x0 for p in wp]), par, max_lp_solves=2)
parts = [prs[g::groups] for g in range(groups)]
for rep in range(2):
    t0 = time.perf_counter()
    with ThreadPoolExecutor(groups) as pool:
        outs = list(pool.map(lambda g: batch.solve_batch_lockstep(parts[g], par, slots, batch=hbs[g]), range(groups)))
    dt = time.perf_counter() - t0
    conv = sum(r.ret == 0 for o in outs for r in o[0])
    print("%d groups x %d slots: %.2f s -> %.2f solves/s; converged %d/%d" % (groups, slots, dt, nsc / dt, conv, nsc), flush=True)
for hb in hbs:
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in hb.stats().items()})
    hb.close()
