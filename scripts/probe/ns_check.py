"""Scratch: one case118-sized normal-phase LP sequence through HIP and oracle, path / iteration / parity per LP."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import activesetmethods_amd as A
from activesetmethods_amd import acopf
from oracle import slp as O
name = sys.argv[1] if len(sys.argv) > 1 else "case118"
nlp = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ls = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
if name.startswith("scen"):
    case = acopf.scenario_case(acopf.synthetic_case("case300", 1, 0.5), int(name[4:]))
else:
    case = acopf.synthetic_case(name, 1, ls)
pr = acopf.acopf_problem(case, name)
mh = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", max_iter=nlp))
t0 = time.time(); sh = A.optimize(mh); th = time.time() - t0
mo = O.Model(pr.n, pr.m, pr.x_L, pr.x_U, pr.g_L, pr.g_U, pr.j_str, pr.eval_f, pr.eval_g, pr.eval_grad_f, pr.eval_jac_g, O.Parameters(algorithm="Line Search", max_iter=nlp))
mo.x[:] = pr.x0
t0 = time.time(); so = O.optimize(mo); to = time.time() - t0
print("hip %.2fs oracle %.2fs" % (th, to))
for k, (a, b) in enumerate(zip(sh.trace, so.trace)):
    sa, sb = a["stats"], b["stats"]
    same = all(np.array_equal(x, y) for x, y in zip(a.get("sets", ()), b.get("sets", ())))
    dp = np.abs(a["p"] - b["p"]).max() / max(1.0, np.abs(b["p"]).max())
    dl = np.abs(a["lam"] - b["lam"]).max() / max(1.0, np.abs(b["lam"]).max())
    if len(sys.argv) > 4 and same and dp < 1e-9 and dl < 1e-9 and sa["path"] == {"warm": 0, "ipm0+ln": 1, "ipm1+ln": 2, "ipm2+ln": 3, "ipm+face": 4, "ipm+ref": 9}.get(sb["path"], -1):
        continue
    print(k, "hip path", sa["path"], "ipm", sa["ipm_iters"], "ns", sa["ns_iters"], "k", sa["ns_dim"], "cold", sa["ns_cold"], "nfact", sa["nfact"], "%.1f ms" % sa["wall_ms"],
          "| oracle", sb["path"], "ipm", sb["ipm_iters"], "ns", sb.get("ns_iters"), "| sets same", same, "dp %.1e dlam %.1e" % (dp, dl))
