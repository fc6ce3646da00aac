"""Host-side mirror of the reference's sub-problem interface (src/algorithms/subproblem.jl) on top of
the C ABI of libasmhip (include/asm_hip.h).

    QpData            subproblem.jl:3-14  (same field names; Q is always None on the SLP path, slp.jl:12)
    HipSubOptimizer   the `AbstractSubOptimizer` (subproblem.jl:1) that replaces `QpModel`; it is created
                      once per SLP run (slp.jl:25-36) and called once per outer iteration with
                      `sub_optimize(x_k, Delta, feasibility)` -> the reference's 6-tuple (subproblem.jl:541)

The Jacobian travels as the COO value vector `dE` in `j_str` order (what `eval_jac_g` fills,
slp.jl:186-191); `compute_jacobian_matrix` (common.jl:12-20) runs on the GPU inside the call.
"""
import ctypes as C
from collections.abc import Mapping
import numpy as np
from . import _lib


class AsmHipError(RuntimeError):
    pass


class PSlack(Mapping):
    """`p_slack::Dict{Int,Vector{Float64}}` of subproblem.jl:495-505 over the flat array the C ABI returned (`raw`: two entries per
    row, NaN where a row has one slack).  The per-row lists are made when a row is first asked for: the SLP drivers hand `raw`
    straight back to the device-side merit function, and a Python loop over 18 637 rows per LP costs milliseconds."""

    def __init__(self, raw, one_slack):
        self.raw, self._one, self._rows = raw, one_slack, None

    def _dict(self):
        if self._rows is None:
            pl, one = self.raw.tolist(), self._one.tolist()
            self._rows = {i: ([pl[2 * i]] if one[i] else [pl[2 * i], pl[2 * i + 1]]) for i in range(len(one))}
        return self._rows

    def __getitem__(self, i):
        if self._rows is None:
            if not 0 <= i < len(self._one):
                raise KeyError(i)
            return [float(self.raw[2 * i])] if self._one[i] else [float(self.raw[2 * i]), float(self.raw[2 * i + 1])]
        return self._rows[i]

    def __iter__(self):
        return iter(range(len(self._one)))

    def __len__(self):
        return len(self._one)

    def __eq__(self, other):
        return self._dict() == (other._dict() if isinstance(other, PSlack) else other)

    def __repr__(self):
        return "PSlack(%r)" % (self._dict(),)


class QpData:
    """LpData(slp) of slp.jl:8-21: c = df, c0 = f, A = Jacobian (as COO values dE), b = E."""

    def __init__(self, c, c0, dE, b, c_lb, c_ub, v_lb, v_ub, sense="MIN_SENSE", Q=None):
        self.sense, self.Q = sense, Q
        self.c, self.c0, self.dE, self.b = c, c0, dE, b
        self.c_lb, self.c_ub, self.v_lb, self.v_ub = c_lb, c_ub, v_lb, v_ub


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class HipSubOptimizer:
    """One handle <-> one HIP stream <-> one LP skeleton (create_model!, subproblem.jl:51-215)."""

    def __init__(self, data, j_row, j_col, device=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        rc = self._lib.asm_create(int(device), C.byref(self._h))
        if rc != 0:
            raise AsmHipError("asm_create(device=%d) failed with code %d (no usable HIP device?)" % (device, rc))
        self.data = data
        self.j_row = np.ascontiguousarray(j_row, dtype=np.int64)
        self.j_col = np.ascontiguousarray(j_col, dtype=np.int64)
        self.n = len(data.v_lb)
        self.m = len(data.c_lb)
        assert self.n > 0 and self.m >= 0                       # subproblem.jl:65-72
        assert len(data.c_ub) == self.m and len(data.v_ub) == self.n
        c_lb, c_ub, v_lb, v_ub = map(_f64, (data.c_lb, data.c_ub, data.v_lb, data.v_ub))
        self._check(self._lib.asm_sublp_setup(self._h, self.n, self.m, len(self.j_row), _lib.i64ptr(self.j_row),
                                              _lib.i64ptr(self.j_col), _lib.dptr(c_lb), _lib.dptr(c_ub),
                                              _lib.dptr(v_lb), _lib.dptr(v_ub)))
        self.nslack = np.where((c_lb > -np.inf) & (c_ub < np.inf), 2, 1)

    def set_bounds(self, data):
        """Re-use the handle (all its HBM buffers, the assembly plan, the device evaluator) for another instance with the same
        pattern and new bounds - the next scenario of a batch."""
        c_lb, c_ub, v_lb, v_ub = map(_f64, (data.c_lb, data.c_ub, data.v_lb, data.v_ub))
        assert len(c_lb) == self.m and len(v_lb) == self.n
        self._check(self._lib.asm_sublp_set_bounds(self._h, _lib.dptr(c_lb), _lib.dptr(c_ub), _lib.dptr(v_lb), _lib.dptr(v_ub)))
        self.data = data
        self.nslack = np.where((c_lb > -np.inf) & (c_ub < np.inf), 2, 1)

    def _check(self, rc):
        if rc != 0:
            raise AsmHipError("libasmhip error %d: %s" % (rc, self._lib.asm_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.asm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ the hot call
    def sub_optimize(self, x_k, Delta, feasibility=False):
        """sub_optimize!(qp, x_k, Δ, feasibility) - subproblem.jl:229-542.
        Returns (Xsol, lambda, mult_x_U, mult_x_L, p_slack, status)."""
        d, n, m = self.data, self.n, self.m
        assert len(d.c) == n and len(d.b) == m and len(x_k) == n   # subproblem.jl:239-246
        self.upload(d.dE, d.c, d.c0, d.b, x_k)
        return self.solve_resident(Delta, feasibility)

    def upload(self, dE, df, f, E, x_k):
        dE, df, E, x_k = map(_f64, (dE, df, E, x_k))
        assert len(dE) == len(self.j_row)
        self._check(self._lib.asm_sublp_upload(self._h, _lib.dptr(dE), _lib.dptr(df), float(f), _lib.dptr(E), _lib.dptr(x_k)))

    def solve_resident(self, Delta, feasibility=False):
        n, m = self.n, self.m
        Xsol = np.empty(n); lam = np.empty(m); mU = np.empty(n); mL = np.empty(n)
        ps = np.empty(2 * max(m, 1)); status = C.c_int32(0)
        self._check(self._lib.asm_sublp_solve_resident(self._h, float(Delta), int(bool(feasibility)), _lib.dptr(Xsol),
                                                       _lib.dptr(lam), _lib.dptr(mU), _lib.dptr(mL), _lib.dptr(ps),
                                                       C.byref(status)))
        p_slack = PSlack(ps, self.nslack == 1)            # subproblem.jl:495-505
        return Xsol, lam, mU, mL, p_slack, int(status.value)

    def lp_solve(self, dE, q, r, lb, ub, w=None, slo=None):
        """The LP as an MOI optimizer receives it (include/asm_hip.h: asm_lp_solve): returns (p, s, y, z, bound_state, status)."""
        dE, q, r, lb, ub = map(_f64, (dE, q, r, lb, ub))
        use = w is not None
        ns = int(self.nslack.sum())
        w_, slo_ = (_f64(w), _f64(slo)) if use else (np.zeros(1), np.zeros(1))
        p = np.empty(self.n); s = np.empty(max(ns, 1)); y = np.empty(max(len(r), 1)); z = np.empty(self.n)
        bs = np.empty(self.n, np.int32); status = C.c_int32(0)
        self._check(self._lib.asm_lp_solve(self._h, _lib.dptr(dE), _lib.dptr(q), _lib.dptr(r), _lib.dptr(lb), _lib.dptr(ub), int(use),
                                           _lib.dptr(w_), _lib.dptr(slo_), _lib.dptr(p), _lib.dptr(s), _lib.dptr(y), _lib.dptr(z),
                                           _lib.i32ptr(bs), C.byref(status)))
        return p, s[:ns], y[:len(r)], z, bs, int(status.value)

    # ------------------------------------------------------------------ observability
    def active_set(self):
        nr, nsl = C.c_int64(0), C.c_int64(0)
        rc = self._lib.asm_sublp_active_set(self._h, None, None, None, C.byref(nr), C.byref(nsl))
        if rc != 0:
            return None
        rows = np.empty(nr.value, np.int32); bnd = np.empty(self.n, np.int32); sl = np.empty(max(nsl.value, 1), np.int32)
        self._check(self._lib.asm_sublp_active_set(self._h, _lib.i32ptr(rows), _lib.i32ptr(bnd), _lib.i32ptr(sl), None, None))
        return rows, bnd, sl[:nsl.value]

    def reset_warm(self):
        self._check(self._lib.asm_sublp_reset_warm(self._h))

    def ns_basis(self):
        """Columns (0-based) retained by the null-space form of the normal phase (asm_sublp_ns_basis); empty when the form is not in use."""
        k = C.c_int64(0)
        self._check(self._lib.asm_sublp_ns_basis(self._h, None, C.byref(k)))
        J = np.zeros(max(int(k.value), 1), np.int32)
        self._check(self._lib.asm_sublp_ns_basis(self._h, J.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(k)))
        return J[:int(k.value)].astype(np.int64)

    def row_order(self):
        """(perm, band, e_rows, e_band): the rows by position in the order of the factorisations (None when the natural order is in use), the
        half-bandwidth, the equality rows of the null-space form in their own order and its band (asm_sublp_row_order)."""
        band, n_e, e_band = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._check(self._lib.asm_sublp_row_order(self._h, None, C.byref(band), None, C.byref(n_e), C.byref(e_band)))
        d = self.data
        M = self.m + int(((d.c_lb > -np.inf) & (d.c_ub < np.inf) & (d.c_lb < d.c_ub)).sum())      # one extra <= row per range constraint
        perm = np.zeros(max(M, 1), np.int32); e = np.zeros(max(int(n_e.value), 1), np.int32)
        self._check(self._lib.asm_sublp_row_order(self._h, perm.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(band),
                                                  e.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n_e), C.byref(e_band)))
        return (perm[:M].astype(np.int64) if band.value > 0 else None), int(band.value), e[:int(n_e.value)].astype(np.int64), int(e_band.value)

    def last_stats(self):
        s = _lib.SolveStats()
        self._check(self._lib.asm_sublp_last_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def kernel_stats(self, reset=False):
        s = _lib.KernelStats()
        self._check(self._lib.asm_kernel_stats_get(self._h, C.byref(s)))
        out = {name: dict(ms=s.ms[i], calls=s.calls[i], flops=s.flops[i], bytes=s.bytes[i]) for i, name in enumerate(_lib.K_NAMES)}
        if reset:
            self._check(self._lib.asm_kernel_stats_reset(self._h))
        return out

    def kernel_timing(self, level):
        """0 off, 1 every launch of the rank-K and panel kernels, 2 every kernel family (asm_kernel_timing)."""
        self._check(self._lib.asm_kernel_timing(self._h, int(level)))

    # ------------------------------------------------------------------ device-side evaluators (include/asm_hip.h: asm_eval_*)
    def eval_setup(self, fm):
        """Hand the flattened function store of a FunctionModel (moi_evaluator.py) and its NLP block kernel to the handle."""
        fl = fm.flatten()
        kind, rows, nnz = 0, 0, 0
        ipar, dpar = np.zeros(1, np.int64), np.zeros(1)
        if fm.nlp is not None:
            if fm.nlp.device is None:
                raise AsmHipError("the model's NLP block has no device kernel")
            name, ipar, dpar = fm.nlp.device
            kind = {"acopf_ohm": 1, "dense_quadratic": 2}[name]
            rows, nnz = fm.nlp.m, len(fm.nlp.rows)
            ipar, dpar = np.ascontiguousarray(ipar, np.int64), np.ascontiguousarray(dpar, np.float64)
        self._ev_keep = (fl, ipar, dpar)
        a = lambda k: fl[k]
        self._check(self._lib.asm_eval_setup(self._h, fl["n_rows"], _lib.i64ptr(a("aff_ptr")), _lib.i64ptr(a("aff_var")), _lib.dptr(a("aff_coef")),
                                             _lib.i64ptr(a("quad_ptr")), _lib.i64ptr(a("q_v1")), _lib.i64ptr(a("q_v2")), _lib.dptr(a("q_coef")),
                                             _lib.dptr(a("constant")), _lib.i64ptr(a("jac_off")), _lib.i64ptr(a("g_ptr")), _lib.i64ptr(a("g_kind")),
                                             _lib.dptr(a("g_coef")), _lib.i64ptr(a("g_other")), float(fl["objective_scale"]), kind, rows, nnz,
                                             _lib.i64ptr(ipar), len(ipar) if kind else 0, _lib.dptr(dpar), len(dpar) if kind else 0))

    def eval_functions(self, x):
        """eval_functions! (slp.jl:186-191) on the GPU: returns (f, df, E); dE stays in HBM as the next LP's input."""
        x = _f64(x)
        f = C.c_double(0.0); df = np.empty(self.n); E = np.empty(max(self.m, 1))
        self._check(self._lib.asm_eval_functions(self._h, _lib.dptr(x), C.byref(f), _lib.dptr(df), _lib.dptr(E)))
        return f.value, df, E[:self.m]

    def eval_constraints(self, x):
        x = _f64(x)
        f = C.c_double(0.0); E = np.empty(max(self.m, 1))
        self._check(self._lib.asm_eval_constraints(self._h, _lib.dptr(x), C.byref(f), _lib.dptr(E)))
        return f.value, E[:self.m]

    def jacobian_values(self):
        dE = np.empty(max(len(self.j_row), 1))
        self._check(self._lib.asm_eval_jacobian_values(self._h, _lib.dptr(dE)))
        return dE[:len(self.j_row)]

    def slp_norms(self, lam, mult_x_U, mult_x_L):
        """(norm_violations(Inf), norm_violations(1), KT_residuals, norm_complementarity(Inf)) - common.jl:35-98 - on the device."""
        out = np.empty(4)
        lam, mult_x_U, mult_x_L = map(_f64, (lam, mult_x_U, mult_x_L))
        self._check(self._lib.asm_slp_norms(self._h, _lib.dptr(lam), _lib.dptr(mult_x_U), _lib.dptr(mult_x_L), _lib.dptr(out)))
        return tuple(float(v) for v in out)

    def slp_merit(self, mode, alpha, p, nu, p_slack, feasibility, prim_infeas):
        """mode 0: compute_phi(x, alpha, p) (slp.jl:79-115); mode 1: compute_derivative (slp.jl:122-147)."""
        ps = self._flat_slacks(p_slack)
        out = C.c_double(0.0)
        p, nu = _f64(p), _f64(nu)
        self._check(self._lib.asm_slp_merit(self._h, int(mode), float(alpha), _lib.dptr(p), _lib.dptr(nu), _lib.dptr(ps), int(bool(feasibility)),
                                            float(prim_infeas) if np.isfinite(prim_infeas) else 0.0, C.byref(out)))
        return out.value

    def _flat_slacks(self, p_slack):
        ps = getattr(p_slack, "raw", None)
        if ps is None:
            ps = np.full(2 * max(self.m, 1), np.nan)
            for i in range(self.m):
                v = p_slack.get(i, [0.0]) if p_slack else [0.0]
                ps[2 * i] = v[0]
                if len(v) > 1:
                    ps[2 * i + 1] = v[1]
        return ps

    def slp_line_search(self, p, nu, p_slack, feasibility, prim_infeas, phi0, D, eta, tau, min_alpha):
        """compute_alpha (slp_line_search.jl:222-244) with the trial points evaluated on the device (asm_slp_line_search).
        Returns (alpha, phi(alpha), trials, ok)."""
        ps = self._flat_slacks(p_slack)
        p, nu = _f64(p), _f64(nu)
        alpha, phi = C.c_double(0.0), C.c_double(0.0)
        trials, ok = C.c_int(0), C.c_int(0)
        self._check(self._lib.asm_slp_line_search(self._h, _lib.dptr(p), _lib.dptr(nu), _lib.dptr(ps), int(bool(feasibility)),
                                                  float(prim_infeas) if np.isfinite(prim_infeas) else 0.0, float(phi0), float(D), float(eta), float(tau),
                                                  float(min_alpha), C.byref(alpha), C.byref(phi), C.byref(trials), C.byref(ok)))
        return alpha.value, phi.value, trials.value, bool(ok.value)

    # per-iteration reductions on the resident Jacobian (common.jl:35-44, slp.jl:54-66)
    def kt_residuals(self, df, lam, mult_x_U, mult_x_L):
        out = C.c_double(0.0)
        df, lam, mult_x_U, mult_x_L = map(_f64, (df, lam, mult_x_U, mult_x_L))
        self._check(self._lib.asm_kt_residuals(self._h, _lib.dptr(df), _lib.dptr(lam), _lib.dptr(mult_x_U), _lib.dptr(mult_x_L), C.byref(out)))
        return out.value

    def jac_row_norms(self):
        out = np.empty(max(self.m, 1))
        self._check(self._lib.asm_jac_row_norms(self._h, _lib.dptr(out)))
        return out[:self.m]
