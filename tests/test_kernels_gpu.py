"""Kernel-level checks of libasmhip against NumPy (float64), through the C ABI test hooks."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle(hip_lib):
    h = C.c_void_p()
    assert hip_lib.asm_create(0, C.byref(h)) == 0
    yield h
    hip_lib.asm_destroy(h)


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("M,K,Ms,tile", [(40, 50, 40, 1), (100, 70, 37, 1), (200, 333, 200, 2), (300, 129, 211, 4),
                                           (500, 1000, 500, 0), (130, 64, 130, 4)])
def test_syrk_matches_numpy(hip_lib, handle, M, K, Ms, tile):
    rng = np.random.default_rng(M * 7 + K)
    A = rng.standard_normal((M, K))
    theta = rng.uniform(0.0, 2.0, K)
    theta[rng.random(K) < 0.2] = 0.0
    idx = np.sort(rng.choice(M, Ms, replace=False)).astype(np.int32)
    diag = rng.uniform(0.0, 1.0, Ms)
    S = np.zeros((Ms, Ms))
    rc = hip_lib.asm_test_syrk(handle, _d(A), M, K, idx.ctypes.data_as(C.POINTER(C.c_int32)), Ms, _d(theta), _d(diag), _d(S), tile)
    assert rc == 0, hip_lib.asm_last_error(handle)
    ref = (A[idx] * theta) @ A[idx].T + np.diag(diag)
    err = np.abs(np.tril(S) - np.tril(ref)).max() / np.abs(ref).max()
    assert err < 1e-13          # f64 MFMA accumulation, K <= 1000


@pytest.mark.parametrize("Ms,K,MsB,srow0,tile", [
    (3100, 1024, -1, 0, 4),        # triangular trailing update, rows not a multiple of the 128-tile: k_syrk_upd, XCD-aware tile order
    (4001, 256, -1, 64, 4),        # odd size, offset origin
    (3333, 256, 768, 128, 4),      # rectangular in-panel update: all rows x the first 768 columns
    (3200, 64, 130, 0, 4),         # shortest k (two ring stages of the update kernel never fill), ragged column count
    (1000, 256, -1, 0, 2), (1000, 256, 300, 64, 2), (300, 64, -1, 0, 1),      # the generic kernel on the small sizes
])
def test_cholesky_update_matches_numpy(hip_lib, handle, Ms, K, MsB, srow0, tile):
    """S -= P P' on the lower triangle (columns < MsB), as the factorisation launches it."""
    rng = np.random.default_rng(Ms + K)
    P = rng.standard_normal((Ms, K))
    S0 = rng.standard_normal((Ms, Ms))
    S = S0.copy()
    rc = hip_lib.asm_test_syrk_update(handle, _d(P), Ms, K, MsB, srow0, _d(S), tile)
    assert rc == 0, hip_lib.asm_last_error(handle)
    ref = S0 - P @ P.T
    mask = np.tril(np.ones((Ms, Ms), bool))
    if MsB >= 0:
        mask[:, MsB:] = False
    scale = np.abs(ref).max()
    assert np.abs(S - ref)[mask].max() / scale < 1e-13
    assert np.array_equal(S[~mask], S0[~mask])          # nothing outside the updated region is touched


@pytest.mark.parametrize("N", [1, 17, 64, 65, 200, 513, 519, 530, 1000, 2500, 4700])      # (519: the thin ninth row tile of the C4 reduced matrix; 530: 18 live rows, padded sub-blocks only)
def test_cholesky_and_solve(hip_lib, handle, N):
    rng = np.random.default_rng(N)
    B = rng.standard_normal((N, N + 5))
    S = B @ B.T + 0.1 * np.eye(N)
    L = np.zeros((N, N))
    assert hip_lib.asm_test_cholesky(handle, _d(S), N, _d(L)) == 0, hip_lib.asm_last_error(handle)
    Lref = np.linalg.cholesky(S)
    assert np.abs(L - Lref).max() / np.abs(Lref).max() < 1e-11
    b = rng.standard_normal(N)
    x = np.zeros(N)
    assert hip_lib.asm_test_chol_solve(handle, _d(S), N, _d(b), _d(x)) == 0
    xref = np.linalg.solve(S, b)
    assert np.abs(x - xref).max() / np.abs(xref).max() < 1e-9
    assert np.abs(S @ x - b).max() < 1e-10 * max(1.0, np.abs(S).max() * np.abs(x).max())


def test_cholesky_pivot_guard(hip_lib, handle):
    """A duplicated row makes S singular: the static guard drops it instead of producing NaNs."""
    rng = np.random.default_rng(3)
    B = rng.standard_normal((30, 50))
    B[7] = B[3]
    S = B @ B.T
    L = np.zeros((30, 30))
    assert hip_lib.asm_test_cholesky(handle, _d(S), 30, _d(L)) == 0
    assert np.isfinite(L).all()
    assert L[7, 7] > 1e100


@pytest.mark.parametrize("M,K", [(5, 3), (130, 77), (500, 1000), (1000, 129)])
def test_gemv(hip_lib, handle, M, K):
    rng = np.random.default_rng(M + K)
    A = rng.standard_normal((M, K)); x = rng.standard_normal(K); y = rng.standard_normal(M)
    Ax = np.zeros(M); ATy = np.zeros(K)
    assert hip_lib.asm_test_gemv(handle, _d(A), M, K, _d(x), _d(y), _d(Ax), _d(ATy)) == 0
    assert np.abs(Ax - A @ x).max() < 1e-12 * K
    assert np.abs(ATy - A.T @ y).max() < 1e-12 * M


@pytest.mark.parametrize("seed,n,m,density,dup,nrange", [(1, 7, 5, 0.5, 0.5, 2), (2, 60, 40, 0.1, 0.3, 5), (3, 50, 30, 1.0, 0.0, 0),
                                                         (4, 33, 21, 0.3, 1.0, 3)])
def test_assembly_bit_exact(hip_lib, seed, n, m, density, dup, nrange):
    """Duplicate accumulation in j_str order is bitwise identical to the oracle's restatement of
    common.jl:12-20, including the stale-entry rule of the extra range rows (subproblem.jl:448-457)."""
    from tests.util import random_subproblem
    from activesetmethods_amd.subproblem import QpData, HipSubOptimizer
    from oracle.subproblem import QpData as OQpData, QpModel, compute_jacobian_matrix
    sp = random_subproblem(seed, n, m, density, dup, nrange)
    opt = HipSubOptimizer(QpData(sp['df'], sp['f'], sp['dE'], sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub']),
                          sp['j_row'], sp['j_col'])
    qp = None
    rng = np.random.default_rng(seed + 100)
    for call in range(3):
        dE = sp['dE'].copy()
        if call == 1:
            dE[rng.random(len(dE)) < 0.4] = 0.0          # entries vanish: stale coefficients must persist
        if call == 2:
            dE = rng.standard_normal(len(dE))
        A, stored = compute_jacobian_matrix(m, n, sp['j_row'] - 1, sp['j_col'] - 1, dE)
        data = OQpData(sp['df'], sp['f'], A, sp['E'], sp['c_lb'], sp['c_ub'], sp['v_lb'], sp['v_ub'], stored)
        if qp is None:
            qp = QpModel(data, sp['j_row'], sp['j_col'])
        qp.data = data
        lp = qp.build_lp(sp['x_k'], 0.4, False)
        M = m + len(qp.adj)
        Jg = np.zeros((M, n))
        assert hip_lib.asm_test_assemble(opt._h, _d(np.ascontiguousarray(dE)), _d(Jg)) == 0
        assert np.array_equal(Jg, lp.A)
    opt.close()


def test_mfma_f64_probe_runs(hip_lib, handle):
    """The FP64 matrix-core probe used for the roofline discussion (profiles/r01_mfma_f64_probe.txt)."""
    t = C.c_double(0.0)
    assert hip_lib.asm_test_mfma_peak(handle, 20000, 2, C.byref(t)) == 0
    assert 5.0 < t.value < 200.0


# ----------------------------------------------------------------------------- round 3: kernels of the null-space form
@pytest.mark.parametrize("Ma,Mb,K,mode", [(64, 64, 32, 0), (130, 200, 96, 0), (130, 200, 96, 1), (519, 1024, 1024, 1), (37, 469, 480, 0)])
def test_gemm_nt_matches_numpy(hip_lib, handle, Ma, Mb, K, mode, monkeypatch):
    """C = (C0) -/+ A B' on the matrix cores (k_gemm_nt and its 32 x 64 / 32 x 96 tile variants): ragged tile edges, both modes, in-place
    accumulation; the variants give the same bits (every entry is the same sum in the same order)."""
    rng = np.random.default_rng(Ma + Mb + K)
    A = rng.standard_normal((Ma, K)); B = rng.standard_normal((Mb, K)); C0 = rng.standard_normal((Ma, Mb))
    ref = A @ B.T if mode == 0 else C0 - A @ B.T
    outs = []
    for variant in ("64", "32", "32w"):
        monkeypatch.setenv("ASM_TEST_GEMM", variant)
        Cout = np.zeros((Ma, Mb))
        rc = hip_lib.asm_test_gemm_nt(handle, _d(A), _d(B), _d(C0), Ma, Mb, K, mode, _d(Cout))
        assert rc == 0, hip_lib.asm_last_error(handle)
        assert np.abs(Cout - ref).max() / np.abs(ref).max() < 1e-13, variant
        outs.append(Cout)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("N,nrhs,backward", [(100, 5, 1), (700, 37, 0), (700, 37, 1), (1500, 64, 1), (2600, 130, 1), (981, 107, 1), (3000, 520, 1)])      # (the last: 32 x 96 tiles, k_gemm_nt32w)
def test_multi_rhs_triangular_solves(hip_lib, handle, N, nrhs, backward):
    """Rows of R solved against the Cholesky factor by right-looking block substitution (products with the explicit inverses of
    the wide diagonal blocks + one update of all remaining columns per block): forward only and forward + backward."""
    from scipy.linalg import solve_triangular
    rng = np.random.default_rng(N + nrhs)
    Bm = rng.standard_normal((N, N + 5))
    S = Bm @ Bm.T + 0.1 * np.eye(N)
    R = rng.standard_normal((nrhs, N))
    X = np.zeros((nrhs, N))
    rc = hip_lib.asm_test_trsm_rows(handle, _d(S), N, _d(R), nrhs, backward, _d(X))
    assert rc == 0, hip_lib.asm_last_error(handle)
    L = np.linalg.cholesky(S)
    ref = solve_triangular(L, R.T, lower=True).T
    if backward:
        ref = solve_triangular(L.T, ref.T, lower=False).T
    assert np.abs(X - ref).max() / np.abs(ref).max() < 1e-9
    if backward:
        assert np.abs(X @ S - R).max() < 1e-9 * max(1.0, np.abs(S).max() * np.abs(X).max())


@pytest.mark.parametrize("N", [11192, 18637])
def test_cholesky_and_solve_at_case1354_sizes(hip_lib, handle, N):
    """The three-level blocked factorisation with look-ahead at the sizes the case1354pegase-sized LPs factor (n = 11192, M = 18637):
    ||L (L'v) - S v|| / (||S|| ||v||) <= 1e-12 for several v, the solve's residual, and at N = 11192 the factor against LAPACK."""
    rng = np.random.default_rng(N)
    Bm = rng.standard_normal((N, 96))
    S = Bm @ Bm.T
    S[np.arange(N), np.arange(N)] += rng.uniform(1.0, 3.0, N) * 96.0
    L = np.zeros((N, N))
    assert hip_lib.asm_test_cholesky(handle, _d(S), N, _d(L)) == 0, hip_lib.asm_last_error(handle)
    assert np.isfinite(L).all() and (np.diag(L) > 0).all() and (np.diag(L) < 1e100).all()
    nS = np.abs(S).sum(axis=1).max()
    for t in range(4):
        v = rng.standard_normal(N) if t else np.ones(N)
        r = L @ (L.T @ v) - S @ v
        assert np.abs(r).max() <= 1e-12 * nS * np.abs(v).max(), (t, np.abs(r).max() / (nS * np.abs(v).max()))
    if N <= 12000:
        Lref = np.linalg.cholesky(S)
        assert np.abs(L - Lref).max() / np.abs(Lref).max() < 1e-11
    del L
    b = rng.standard_normal(N)
    x = np.zeros(N)
    assert hip_lib.asm_test_chol_solve(handle, _d(S), N, _d(b), _d(x)) == 0, hip_lib.asm_last_error(handle)
    assert np.abs(S @ x - b).max() <= 1e-11 * max(1.0, nS * np.abs(x).max())


def test_panel_wait_timeout_is_reported_once_and_the_handle_survives(hip_lib, handle):
    """The dataflow panel kernel's bounded wait with a producer that never publishes: every waiting workgroup gives up (the first after
    the full bound, the others at their next look at the timeout word), the host reports ASM_ERR_HIP with the kernel's message once,
    and the same handle factors correctly afterwards."""
    import time
    t0 = time.time()
    rc = hip_lib.asm_test_panel_timeout(handle, 8)
    dt = time.time() - t0
    assert rc == -2, rc                                         # ASM_ERR_HIP
    assert b"timed out" in hip_lib.asm_last_error(handle)
    assert dt < 60.0
    N = 300
    rng = np.random.default_rng(5)
    Bm = rng.standard_normal((N, N + 5))
    S = Bm @ Bm.T + 0.1 * np.eye(N)
    L = np.zeros((N, N))
    assert hip_lib.asm_test_cholesky(handle, _d(S), N, _d(L)) == 0, hip_lib.asm_last_error(handle)      # no stale timeout
    assert np.abs(L - np.linalg.cholesky(S)).max() < 1e-10


@pytest.mark.parametrize("N,band,nrhs", [(700, 90, 33), (2300, 267, 70), (3100, 1398, 130), (5000, 150, 40), (2245, 40, 137)])
def test_banded_cholesky_and_solves(hip_lib, handle, N, band, nrhs):
    """A banded SPD matrix (the S0 = A_EF A_EF' of the null-space form in reverse Cuthill-McKee order): with the band declared, the
    factorisation, the single right-hand-side solve and the multi right-hand-side block substitution stop at the band - same factor
    and solutions as the dense NumPy reference."""
    from scipy.linalg import solve_triangular
    rng = np.random.default_rng(N + band)
    Bm = np.zeros((N, N))
    for d in range(0, band // 2 + 1):                 # B has half-bandwidth band // 2 -> B B' has half-bandwidth <= band
        v = rng.standard_normal(N - d)
        Bm[np.arange(d, N), np.arange(0, N - d)] = v
    S = Bm @ Bm.T + 0.5 * np.eye(N)
    i, j = np.nonzero(S)
    assert np.abs(i - j).max() <= band
    assert hip_lib.asm_test_set_band(handle, band) == 0
    try:
        L = np.zeros((N, N))
        assert hip_lib.asm_test_cholesky(handle, _d(S), N, _d(L)) == 0, hip_lib.asm_last_error(handle)
        Lref = np.linalg.cholesky(S)
        assert np.abs(L - Lref).max() / np.abs(Lref).max() < 1e-11
        b = rng.standard_normal(N)
        x = np.zeros(N)
        assert hip_lib.asm_test_chol_solve(handle, _d(S), N, _d(b), _d(x)) == 0
        assert np.abs(S @ x - b).max() < 1e-10 * max(1.0, np.abs(S).max() * np.abs(x).max())
        R = rng.standard_normal((nrhs, N))
        for backward in (0, 1):
            X = np.zeros((nrhs, N))
            assert hip_lib.asm_test_trsm_rows(handle, _d(S), N, _d(R), nrhs, backward, _d(X)) == 0, hip_lib.asm_last_error(handle)
            ref = solve_triangular(Lref, R.T, lower=True).T
            if backward:
                ref = solve_triangular(Lref.T, ref.T, lower=False).T
            assert np.abs(X - ref).max() / np.abs(ref).max() < 1e-9
    finally:
        assert hip_lib.asm_test_set_band(handle, 0) == 0
