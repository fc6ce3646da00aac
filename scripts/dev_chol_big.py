import sys, time, ctypes as C; sys.path.insert(0,'.')
import numpy as np
from activesetmethods_amd import _lib
lib=_lib.load(); h=C.c_void_p(); assert lib.asm_create(0,C.byref(h))==0
d=lambda a:a.ctypes.data_as(C.POINTER(C.c_double))
for N in [int(a) for a in sys.argv[1:]]:
    rng=np.random.default_rng(N)
    B=rng.standard_normal((N,N//4+7)); S=B@B.T+np.diag(rng.uniform(0.5,2.0,N))
    L=np.zeros((N,N)); lib.asm_kernel_stats_reset(h); t=time.time(); rc=lib.asm_test_cholesky(h,d(S),N,d(L)); t=time.time()-t
    ks=_lib.KernelStats(); lib.asm_kernel_stats_get(h,C.byref(ks)); print('chol device ms',ks.ms[4],'TFLOP/s',ks.flops[4]/ks.ms[4]/1e9)
    assert rc==0, lib.asm_last_error(h)
    Lr=np.linalg.cholesky(S)
    err=np.abs(L-Lr).max()/np.abs(Lr).max()
    rows=np.abs(L-Lr).max(axis=1); bad=np.argsort(-rows)[:5]
    b=rng.standard_normal(N); x=np.zeros(N); lib.asm_test_chol_solve(h,d(S),N,d(b),d(x))
    print(N,'chol rel err',err,'worst rows',bad,rows[bad],'solve resid',np.abs(S@x-b).max(),'time',round(t,2))
