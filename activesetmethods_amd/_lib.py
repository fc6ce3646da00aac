"""ctypes binding of libasmhip.so (include/asm_hip.h).  The HIP library is the product: loading fails
loudly when it is missing - there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("ASM_HIP_LIB", "libasmhip.so"))      # ASM_HIP_LIB: another build in the package directory (A/B timing in one GPU call)

OPTIMAL, INFEASIBLE, DUAL_INFEASIBLE, OTHER = 1, 2, 3, 4
K_NAMES = ("assemble", "scale", "gemv", "syrk", "chol", "trsv", "syrk_kernel", "panel_kernel")


class SolveStats(C.Structure):
    _fields_ = [("path", C.c_int32), ("polished", C.c_int32), ("ipm_iters", C.c_int32), ("nfact", C.c_int32),
                ("eqp", C.c_int32), ("M", C.c_int32), ("n", C.c_int32), ("ns", C.c_int32), ("col_iters", C.c_int32),
                ("ns_iters", C.c_int32), ("ns_dim", C.c_int32), ("ns_cold", C.c_int32), ("restored", C.c_int32),
                ("ipm_pinf", C.c_double), ("ipm_dinf", C.c_double), ("ipm_gap", C.c_double),
                ("kkt_pr", C.c_double), ("kkt_du", C.c_double), ("wall_ms", C.c_double)]


class KernelStats(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("calls", C.c_int64 * 8), ("flops", C.c_double * 8), ("bytes", C.c_double * 8)]


class SlpParams(C.Structure):
    """asm_slp_params (src/parameters.jl:17-28)."""
    _fields_ = [("max_iter", C.c_int32), ("max_lp_solves", C.c_int32), ("tol_direction", C.c_double), ("tol_residual", C.c_double),
                ("tol_infeas", C.c_double), ("eta", C.c_double), ("tau", C.c_double), ("min_alpha", C.c_double)]


class SlpResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("iter", C.c_int32), ("lp_solves", C.c_int32), ("restoration_solves", C.c_int32),
                ("ls_trials", C.c_int32), ("slot", C.c_int32), ("paths", C.c_int32 * 12), ("ipm_iters", C.c_int32), ("ns_cold", C.c_int32),
                ("obj_val", C.c_double), ("prim_infeas", C.c_double), ("dual_infeas", C.c_double), ("compl_", C.c_double)]


class BatchStats(C.Structure):
    _fields_ = [("rounds", C.c_int64), ("ops", C.c_int64), ("launches", C.c_int64), ("releases", C.c_int64), ("blob_bytes", C.c_int64),
                ("emit_ms", C.c_double), ("wait_ms", C.c_double), ("host_ms", C.c_double), ("wall_ms", C.c_double),
                ("panel_ms", C.c_double), ("panel_launches", C.c_int64), ("panel_ops", C.c_int64), ("panel_flops", C.c_double), ("panel_bytes", C.c_double)]


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I64 = C.POINTER(C.c_int64)
_I32 = C.POINTER(C.c_int32)

PROTOTYPES = {
    "asm_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "asm_destroy": (C.c_int, [_P]),
    "asm_last_error": (C.c_char_p, [_P]),
    "asm_sublp_setup": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _I64, _I64, _D, _D, _D, _D]),
    "asm_sublp_set_bounds": (C.c_int, [_P, _D, _D, _D, _D]),
    "asm_sublp_solve": (C.c_int, [_P, _D, _D, C.c_double, _D, _D, C.c_double, C.c_int, _D, _D, _D, _D, _D, _I32]),
    "asm_lp_solve": (C.c_int, [_P, _D, _D, _D, _D, _D, C.c_int, _D, _D, _D, _D, _D, _D, _I32, _I32]),
    "asm_sublp_upload": (C.c_int, [_P, _D, _D, C.c_double, _D, _D]),
    "asm_sublp_solve_resident": (C.c_int, [_P, C.c_double, C.c_int, _D, _D, _D, _D, _D, _I32]),
    "asm_sublp_active_set": (C.c_int, [_P, _I32, _I32, _I32, _I64, _I64]),
    "asm_sublp_reset_warm": (C.c_int, [_P]),
    "asm_sublp_ns_basis": (C.c_int, [_P, _I32, _I64]),
    "asm_sublp_row_order": (C.c_int, [_P, _I32, _I64, _I32, _I64, _I64]),
    "asm_sublp_last_stats": (C.c_int, [_P, C.POINTER(SolveStats)]),
    "asm_kernel_stats_get": (C.c_int, [_P, C.POINTER(KernelStats)]),
    "asm_kernel_stats_reset": (C.c_int, [_P]),
    "asm_kernel_timing": (C.c_int, [_P, C.c_int]),
    "asm_kt_residuals": (C.c_int, [_P, _D, _D, _D, _D, _D]),
    "asm_jac_row_norms": (C.c_int, [_P, _D]),
    "asm_eval_setup": (C.c_int, [_P, C.c_int64, _I64, _I64, _D, _I64, _I64, _I64, _D, _D, _I64, _I64, _I64, _D, _I64, C.c_double, C.c_int,
                                 C.c_int64, C.c_int64, _I64, C.c_int64, _D, C.c_int64]),
    "asm_eval_functions": (C.c_int, [_P, _D, _D, _D, _D]),
    "asm_eval_constraints": (C.c_int, [_P, _D, _D, _D]),
    "asm_eval_jacobian_values": (C.c_int, [_P, _D]),
    "asm_slp_norms": (C.c_int, [_P, _D, _D, _D, _D]),
    "asm_slp_merit": (C.c_int, [_P, C.c_int, C.c_double, _D, _D, _D, C.c_int, C.c_double, _D]),
    "asm_slp_line_search": (C.c_int, [_P, _D, _D, _D, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _D, _D,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "asm_sublp_set_ns_basis": (C.c_int, [_P, _I32, C.c_int64]),
    "asm_slp_run": (C.c_int, [_P, C.POINTER(SlpParams), _D, _D, _D, _D, _D, _D, C.POINTER(SlpResult)]),
    "asm_batch_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "asm_batch_destroy": (C.c_int, [_P]),
    "asm_batch_last_error": (C.c_char_p, [_P]),
    "asm_batch_slots": (C.c_int, [_P]),
    "asm_batch_set_groups": (C.c_int, [_P, C.c_int]),
    "asm_batch_groups": (C.c_int, [_P]),
    "asm_batch_handle": (_P, [_P, C.c_int]),
    "asm_batch_setup": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _I64, _I64, _D, _D, _D, _D]),
    "asm_batch_eval_setup": (C.c_int, [_P, C.c_int64, _I64, _I64, _D, _I64, _I64, _I64, _D, _D, _I64, _I64, _I64, _D, _I64, C.c_double, C.c_int,
                                       C.c_int64, C.c_int64, _I64, C.c_int64, _D, C.c_int64]),
    "asm_batch_set_ns_basis": (C.c_int, [_P, _I32, C.c_int64]),
    "asm_batch_ns_basis": (C.c_int, [_P, _I32, _I64]),
    "asm_batch_sublp_solve": (C.c_int, [_P, C.c_int, _D, _D, _D, _D, _D, _D, _D, _D, _D, _D, _I32, _D, _D, _D, _D, _D, _I32]),
    "asm_batch_slp_run": (C.c_int, [_P, C.c_int64, _D, _D, _D, _D, _D, C.POINTER(SlpParams), _D, _D, _D, _D, _D, C.POINTER(SlpResult)]),
    "asm_batch_get_stats": (C.c_int, [_P, C.POINTER(BatchStats)]),
    "asm_test_syrk": (C.c_int, [_P, _D, C.c_int64, C.c_int64, _I32, C.c_int64, _D, _D, _D, C.c_int]),
    "asm_test_syrk_update": (C.c_int, [_P, _D, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _D, C.c_int]),
    "asm_test_cholesky": (C.c_int, [_P, _D, C.c_int64, _D]),
    "asm_test_chol_solve": (C.c_int, [_P, _D, C.c_int64, _D, _D]),
    "asm_test_no_polish": (C.c_int, [_P, C.c_int]),
    "asm_test_panel_timeout": (C.c_int, [_P, C.c_int]),
    "asm_test_set_band": (C.c_int, [_P, C.c_int]),
    "asm_test_gemm_nt": (C.c_int, [_P, _D, _D, _D, C.c_int64, C.c_int64, C.c_int64, C.c_int, _D]),
    "asm_test_trsm_rows": (C.c_int, [_P, _D, C.c_int64, _D, C.c_int64, C.c_int, _D]),
    "asm_test_gemv": (C.c_int, [_P, _D, C.c_int64, C.c_int64, _D, _D, _D, _D]),
    "asm_test_assemble": (C.c_int, [_P, _D, _D]),
    "asm_test_mfma_peak": (C.c_int, [_P, C.c_int, C.c_int, _D]),
}

_lib = None


def load():
    """Load libasmhip.so and bind every symbol include/asm_hip.h declares."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C activesetmethods_amd/csrc).  There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)      # AttributeError here = ABI drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def dptr(a):
    return a.ctypes.data_as(_D)


def i64ptr(a):
    return a.ctypes.data_as(_I64)


def i32ptr(a):
    return a.ctypes.data_as(_I32)
