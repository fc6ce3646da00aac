"""Two banded Cholesky factorisations at the same time on two handles (two streams) against one after the other: do their dependent chains
overlap?  (Feasibility probe of a split S0 factorisation.)  usage: chol_concurrent.py N band"""
import os, sys, time, ctypes as C, threading
sys.path.insert(0, "."); os.environ.setdefault("ASM_HIP_TIMING", "2")
import numpy as np
from activesetmethods_amd import _lib
lib = _lib.load()
N, band = int(sys.argv[1]), int(sys.argv[2])
d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
rng = np.random.default_rng(1)
S = np.zeros((N, N))
for off in range(0, band + 1, 7):
    v = rng.standard_normal(N - off) * 0.1
    S += np.diag(v, -off) + (np.diag(v, off) if off else 0)
S = S + S.T + (np.abs(S).sum(1).max() + 1.0) * np.eye(N)
hs = []
for k in range(2):
    h = C.c_void_p(); assert lib.asm_create(0, C.byref(h)) == 0
    assert lib.asm_test_set_band(h, band) == 0
    hs.append(h)
Ls = [np.zeros((N, N)) for _ in range(2)]
ms = [0.0, 0.0]
def fac(k):
    lib.asm_kernel_stats_reset(hs[k])
    assert lib.asm_test_cholesky(hs[k], d(S), N, d(Ls[k])) == 0, lib.asm_last_error(hs[k])
    ks = _lib.KernelStats(); lib.asm_kernel_stats_get(hs[k], C.byref(ks))
    ms[k] = ks.ms[4]
for k in range(2): fac(k)           # allocate / warm up
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
one = timed(lambda: fac(0)); one_ms = ms[0]
def both():
    th = [threading.Thread(target=fac, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
two = timed(both); two_ms = list(ms)
seq = timed(lambda: (fac(0), fac(1)))
print("factorisation alone (HIP events) %.3f ms; each of two at the same time %.3f / %.3f ms" % (one_ms, two_ms[0], two_ms[1]))
print("N %d band %d: one %.2f ms (includes the host copies of the test hook), two one after the other %.2f ms, two at the same time %.2f ms" % (N, band, one, seq, two))
