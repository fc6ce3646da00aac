import sys, ctypes as C; sys.path.insert(0,'.')
from activesetmethods_amd import _lib
lib=_lib.load(); h=C.c_void_p(); assert lib.asm_create(0,C.byref(h))==0
for wps in (1,2,4,6,8):
    for it in (100000,):
        t=C.c_double(0); rc=lib.asm_test_mfma_peak(h,it,wps,C.byref(t)); print('waves/SIMD',wps,'iters',it,'rc',rc,'FP64 MFMA TFLOP/s %.2f'%t.value)
