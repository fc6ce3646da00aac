"""Cholesky timing through the kernel test hook (development probe; GPU box).  usage: chol_time.py N[:band] [N[:band] ...]"""
import os, sys, ctypes as C; sys.path.insert(0, "."); os.environ.setdefault("ASM_HIP_TIMING", "2")
import numpy as np
from activesetmethods_amd import _lib
if os.environ.get("ASM_LIB"): _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["ASM_LIB"])      # A/B against another build in one call
lib = _lib.load()
for spec in sys.argv[1:]:
    N, band = (int(v) for v in (spec.split(':') + ['0'])[:2])
    h = C.c_void_p(); assert lib.asm_create(0, C.byref(h)) == 0
    rng = np.random.default_rng(N)
    if band:
        B = np.zeros((N, N))
        for dd in range(0, band // 2 + 1, max(1, band // 24)):
            B[np.arange(dd, N), np.arange(0, N - dd)] = rng.standard_normal(N - dd)
        S = B @ B.T + 0.5 * np.eye(N)
        assert lib.asm_test_set_band(h, band) == 0
    else:
        B = rng.standard_normal((N, 64))
        S = B @ B.T + N * np.eye(N)
    L = np.zeros((N, N))
    d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    best = 1e9
    for rep in range(4):
        lib.asm_kernel_stats_reset(h)
        assert lib.asm_test_cholesky(h, d(S), N, d(L)) == 0, lib.asm_last_error(h)
        ks = _lib.KernelStats(); lib.asm_kernel_stats_get(h, C.byref(ks))
        best = min(best, ks.ms[4])
    err = np.abs(np.tril(L @ L.T - S)).max() / np.abs(S).max() if N <= 6000 else float('nan')
    print('N %6d band %5d  chol %8.3f ms  %6.1f us/step  %6.2f TFLOP/s  resid %.1e' % (N, band, best, 1e3 * best / ((N + 63) // 64), N ** 3 / 3 / best / 1e9, err), flush=True)
    lib.asm_destroy(h)
