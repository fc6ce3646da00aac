"""Per-phase cycle sums of k_syrk_upd from the diagnostic build (libasmhip_prof.so: hipcc ... -DASM_UPD_PROF).  Development probe; GPU box.
usage: upd_prof.py N"""
import os, sys, ctypes as C; sys.path.insert(0, "."); os.environ.setdefault("ASM_HIP_TIMING", "0")
import numpy as np
from activesetmethods_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libasmhip_prof.so")
lib = _lib.load()
N = int(sys.argv[1])
h = C.c_void_p(); assert lib.asm_create(0, C.byref(h)) == 0
rng = np.random.default_rng(N)
B = rng.standard_normal((N, 64)); S = B @ B.T + N * np.eye(N); L = np.zeros((N, N))
d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
out = (C.c_ulonglong * 8)()
raw = C.CDLL(_lib.LIB_PATH)
for rep in range(3):
    assert lib.asm_test_cholesky(h, d(S), N, d(L)) == 0
    raw.asm_debug_upd_prof(out)
    v = list(out); ch = max(v[6], 1)
    print("rep %d: workgroups %d | per workgroup (cycles, wave 0): prologue %.0f (S loads issued by %.0f)  loop %.0f (%.0f per chunk)  epilogue %.0f" %
          (rep, v[7], v[0] / max(v[7], 1), v[2] / max(v[7], 1), v[5] / max(v[7], 1), v[5] / ch, v[1] / max(v[7], 1)), flush=True)
