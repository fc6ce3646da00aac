"""Phase stamps of potrf64_body from the diagnostic build (libasmhip_prof.so: hipcc ... -DASM_POTRF_PROF): the kernel prints clock64
deltas for the diagonal block at k0 = 64.  Development probe; GPU box.   usage: potrf_prof.py [N]"""
import os, sys, ctypes as C; sys.path.insert(0, ".")
import numpy as np
from activesetmethods_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libasmhip_prof.so")
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = C.c_void_p(); assert lib.asm_create(0, C.byref(h)) == 0
rng = np.random.default_rng(N)
B = rng.standard_normal((N, 64)); S = B @ B.T + N * np.eye(N); L = np.zeros((N, N))
d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
for rep in range(3):
    assert lib.asm_test_cholesky(h, d(S), N, d(L)) == 0
lib.asm_destroy(h)
