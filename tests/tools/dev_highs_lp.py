"""Times SciPy/HiGHS (sparse dual simplex and IPM) on the first sub-LP of a workload - an independent third-party
CPU LP baseline on the identical LP.  Uses oracle/ only to build the LP the reference would pose."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, scipy.sparse as sp
from scipy.optimize import linprog
from activesetmethods_amd import acopf, problems
from oracle import slp as O
from oracle.subproblem import QpData, QpModel

def first_lp(pr, alg):
    from oracle.subproblem import compute_jacobian_matrix
    x = pr.x0.copy()
    dE = pr.eval_jac_g(x, np.zeros(pr.nnz))
    A, st = compute_jacobian_matrix(pr.m, pr.n, pr.j_row - 1, pr.j_col - 1, dE)
    qp = QpModel(QpData(pr.eval_grad_f(x, np.zeros(pr.n)), pr.eval_f(x), A, pr.eval_g(x, np.zeros(pr.m)), pr.g_L, pr.g_U, pr.x_L, pr.x_U, st),
                 pr.j_row, pr.j_col)
    return qp.build_lp(x, 1000.0 if alg == "Line Search" else 0.4, False)

def highs(lp, method):
    A = sp.csr_matrix(lp.A)
    eq = lp.rtype == 0; ge = lp.rtype == 1; le = lp.rtype == -1
    A_ub = sp.vstack([-A[ge], A[le]]).tocsr(); b_ub = np.concatenate([-lp.r[ge], lp.r[le]])
    t = time.time()
    res = linprog(lp.q, A_ub=A_ub, b_ub=b_ub, A_eq=A[eq], b_eq=lp.r[eq], bounds=np.c_[lp.lb, lp.ub], method=method)
    return time.time() - t, res

if __name__ == "__main__":
    name = sys.argv[1]; alg = sys.argv[2] if len(sys.argv) > 2 else "Line Search"
    pr = acopf.acopf_problem(acopf.synthetic_case(name, 1), name) if name.startswith("case") else problems.synthetic_dense_nlp()
    lp = first_lp(pr, alg)
    print("LP", lp.A.shape, "nnz", int((lp.A != 0).sum()))
    for method in ("highs-ds", "highs-ipm"):
        dt, res = highs(lp, method)
        print(method, "status", res.status, "fun", res.fun, "nit", res.nit, "time %.3f s" % dt)
