"""What a nested-dissection order of S0 = A_EF A_EF' would buy (development probe, CPU): the case1354pegase-sized grid, reverse Cuthill-McKee band of
the whole pattern against a spectral bisection (vertex separator + RCM band of the two halves).  Result (round 4): band 1397 over 10 673 rows =
168 sequential 64-wide steps; bisection: separator 474 rows, halves 4 862 / 5 337 rows with bands 905 / 1 005 = 84 + 8 steps when the halves run
side by side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
from activesetmethods_amd import acopf
case = acopf.synthetic_case("case1354pegase", 1, 0.5)
pr = acopf.acopf_problem(case, "c4")
x = pr.x0.copy()
dE = pr.eval_jac_g(x, np.zeros(pr.nnz))
J = sp.csr_matrix((np.abs(dE) + 1e-300, (pr.j_row - 1, pr.j_col - 1)), shape=(pr.m, pr.n))
eq = pr.g_L == pr.g_U
free = pr.x_U > pr.x_L
A = J[eq][:, free]
print("A_EF", A.shape, "nnz", A.nnz)
S = (A @ A.T).tocsr(); S.data[:] = 1.0
n = S.shape[0]
perm = reverse_cuthill_mckee(S, symmetric_mode=True)
Sp = S[perm][:, perm].tocoo()
print("RCM band", int(np.abs(Sp.row - Sp.col).max()))
# level-structure bisection along the RCM order: separator = rows within `band` of the cut ... count rows coupling both halves
for cut_frac in (0.5,):
    cut = int(n * cut_frac)
    left = np.zeros(n, bool); left[:cut] = True
    r, c = Sp.row, Sp.col
    cross = (left[r] != left[c])
    sep_rows = np.unique(np.minimum(r[cross], c[cross]) * 0 + np.where(left[r[cross]], r[cross], c[cross]))   # left endpoints of crossing edges
    print("cut at", cut, ": crossing entries", int(cross.sum()), " left-side separator rows", len(sep_rows))
# vertex separator via spectral bisection of the row graph
from scipy.sparse.linalg import eigsh
L = sp.csgraph.laplacian(S, normed=False).astype(float)
t0 = time.time()
try:
    w, v = eigsh(L, k=2, sigma=-1e-3, which="LM")
    f = v[:, 1]
    side = f > np.median(f)
    r, c = S.tocoo().row, S.tocoo().col
    cross = side[r] != side[c]
    cand = np.unique(r[cross & side[r]])
    print("spectral bisection: %d / %d rows, crossing entries %d, one-sided vertex separator %d rows (%.1f s)" % (side.sum(), (~side).sum(), int(cross.sum()), len(cand), time.time() - t0))
    # band of each half after removing the separator
    keep = np.ones(n, bool); keep[cand] = False
    for nm, msk in (("A", side & keep), ("B", (~side) & keep)):
        idx = np.nonzero(msk)[0]
        Sh = S[idx][:, idx]
        p2 = reverse_cuthill_mckee(Sh, symmetric_mode=True)
        Sh2 = Sh[p2][:, p2].tocoo()
        print("  half", nm, len(idx), "rows, RCM band", int(np.abs(Sh2.row - Sh2.col).max()) if Sh2.nnz else 0)
except Exception as e:
    print("spectral failed", e)
