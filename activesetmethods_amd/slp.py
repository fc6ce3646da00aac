"""Host-side callers of the HIP sub-problem path: the reference's `Model` container and its two SLP
outer loops, restated in Python so that the hot path can be driven where no Julia runtime exists
(SURVEY.md section 8 row f2).  Same names, parameters, status codes and control flow as the reference:

    Model, optimize        src/model.jl:1-80
    SlpLS.run              src/algorithms/slp_line_search.jl:78-261
    SlpTR.run              src/algorithms/slp_trust_region.jl:87-251
    merit / norm helpers   src/algorithms/slp.jl:54-147, src/algorithms/common.jl:35-98

Every LP sub-problem goes through `parameters.external_optimizer` (default: HipSubOptimizer ->
libasmhip); the Jacobian-consuming reductions (`KT_residuals`, the row norms of `compute_nu!`) are
evaluated on the Jacobian already resident in HBM through the same handle.
"""
import time

import numpy as np

from . import _lib
from .parameters import Parameters
from .subproblem import HipSubOptimizer, QpData

INF = np.inf
OPTIMAL, INFEASIBLE = _lib.OPTIMAL, _lib.INFEASIBLE


class Model:
    """src/model.jl:1-61."""

    def __init__(self, n, m, x_L, x_U, g_L, g_U, j_row, j_col, eval_f, eval_g, eval_grad_f, eval_jac_g, parameters=None):
        self.n, self.m = int(n), int(m)
        self.x = np.zeros(n)
        self.x_L, self.x_U = np.asarray(x_L, float), np.asarray(x_U, float)
        self.g = np.zeros(m)
        self.g_L, self.g_U = np.asarray(g_L, float), np.asarray(g_U, float)
        self.j_row, self.j_col = np.asarray(j_row, np.int64), np.asarray(j_col, np.int64)
        self.mult_g, self.mult_x_L, self.mult_x_U = np.zeros(m), np.zeros(n), np.zeros(n)
        self.obj_val = 0.0
        self.status = -5
        self.eval_f, self.eval_g, self.eval_grad_f, self.eval_jac_g = eval_f, eval_g, eval_grad_f, eval_jac_g
        self.parameters = parameters or Parameters()
        self.statistics = {}

    @classmethod
    def from_problem(cls, pr, parameters=None):
        mdl = cls(pr.n, pr.m, pr.x_L, pr.x_U, pr.g_L, pr.g_U, pr.j_row, pr.j_col, pr.eval_f, pr.eval_g, pr.eval_grad_f,
                  pr.eval_jac_g, parameters)
        mdl.x[:] = pr.x0
        mdl.function_model = getattr(pr, "function_model", None)      # the device evaluator's input (Parameters.device_eval)
        return mdl


def _default_factory(data, j_row, j_col):
    return HipSubOptimizer(data, j_row, j_col)


class AbstractSlpOptimizer:
    """State shared by both algorithms (slp_line_search.jl:4-71, slp_trust_region.jl:10-80)."""

    def __init__(self, problem):
        n, m = problem.n, problem.m
        self.problem = problem
        self.options = problem.parameters
        self.x = np.zeros(n); self.p = np.zeros(n); self.p_slack = {}
        self.lam = np.zeros(m); self.mult_x_L = np.zeros(n); self.mult_x_U = np.zeros(n)
        self.f = 0.0; self.df = np.zeros(n); self.E = np.zeros(m); self.dE = np.zeros(len(problem.j_row))
        self.phi = INF; self.nu = np.zeros(m)
        self.prim_infeas = self.dual_infeas = self.compl = INF
        self.optimizer = None
        self.feasibility_restoration = False
        self.iter, self.ret = 1, -5
        self.lp_solves = 0
        self.lp_time = 0.0
        self.trace = []
        self._uploaded = False
        # device-side evaluation / reductions (Parameters.device_eval): needs the FunctionModel the problem was built from
        self._fm = getattr(problem, "function_model", None) if getattr(problem.parameters, "device_eval", False) else None
        if getattr(problem.parameters, "device_eval", False) and self._fm is None:
            raise ValueError("device_eval needs a problem built from a FunctionModel (activesetmethods_amd.moi_evaluator)")
        self._norms = None
        # masks used by the merit function (slp.jl:90-96)
        gl, gu = problem.g_L, problem.g_U
        self._both = (gl > -INF) & (gu < INF)
        self._lo = (gl > -INF) & ~self._both
        self._up = (gu < INF) & ~self._both & ~(gl > -INF)

    # slp.jl:186-191
    def eval_functions(self):
        pr = self.problem
        self._norms = None
        if self._fm is not None:                          # f, df, E come back; the Jacobian values stay in HBM
            self._ensure_optimizer(upload=False)
            self.f, self.df, self.E = self.optimizer.eval_functions(self.x)
            self._x_dev = self.x.copy()                   # the iterate the device evaluator holds
            self._uploaded = True
            return
        self.f = pr.eval_f(self.x)
        pr.eval_grad_f(self.x, self.df)
        pr.eval_g(self.x, self.E)
        pr.eval_jac_g(self.x, self.dE)
        self._uploaded = False

    def _ensure_optimizer(self, upload=True):
        pr = self.problem
        data = QpData(self.df, self.f, self.dE, self.E, pr.g_L, pr.g_U, pr.x_L, pr.x_U)      # LpData, slp.jl:8-21
        if self.optimizer is None:                                                          # slp.jl:24-36
            factory = self.options.external_optimizer or _default_factory
            self.optimizer = factory(data, pr.j_row, pr.j_col)
            if self._fm is not None:
                self.optimizer.eval_setup(self._fm)
        else:
            self.optimizer.data = data                                                      # slp.jl:38-40
        if upload and not self._uploaded:
            self.optimizer.upload(self.dE, self.df, self.f, self.E, self.x)
            self._uploaded = True

    # slp.jl:23-47
    def sub_optimize(self, Delta=1000.0):
        self._ensure_optimizer()
        t0 = time.perf_counter()
        out = self.optimizer.solve_resident(Delta, self.feasibility_restoration)
        self.lp_time += time.perf_counter() - t0
        self.lp_solves += 1
        st = self.optimizer.last_stats()
        rec = dict(iter=self.iter, fr=bool(self.feasibility_restoration), status=out[5], delta=float(Delta), x=self.x.copy(),
                   p=out[0].copy(), lam=out[1].copy(), mult_x_U=out[2].copy(), mult_x_L=out[3].copy(), stats=st)
        if self.feasibility_restoration:
            rec['p_slack'] = out[4]                         # slack values of the restoration LP (subproblem.jl:531-541)
        if out[5] == OPTIMAL:
            rec['sets'] = self.optimizer.active_set()
        self.trace.append(rec)
        return out

    def _device_norms(self):
        key = (self.lam.tobytes(), self.mult_x_U.tobytes(), self.mult_x_L.tobytes())
        if self._norms is None or self._norms[0] != key:
            self._norms = (key, self.optimizer.slp_norms(self.lam, self.mult_x_U, self.mult_x_L))
        return self._norms[1]

    # common.jl:35-44 on the HBM-resident Jacobian
    def KT_residuals(self):
        if self._fm is not None:
            return self._device_norms()[2]
        self._ensure_optimizer()
        return self.optimizer.kt_residuals(self.df, self.lam, self.mult_x_U, self.mult_x_L)

    # common.jl:51-68
    def norm_complementarity(self, p=INF):
        pr = self.problem
        if self._fm is not None and p == INF:
            return self._device_norms()[3]
        ineq = pr.g_L != pr.g_U
        compl = np.where(ineq, np.minimum(self.E - pr.g_L, pr.g_U - self.E) * self.lam, 0.0)
        denom = float(np.sum(self.lam[ineq] ** 2))
        nrm = np.linalg.norm(compl, p) if pr.m else 0.0
        return nrm / (1 + np.sqrt(denom))

    # common.jl:75-98
    def norm_violations(self, p=1, E=None):
        pr = self.problem
        if self._fm is not None and E is None and p in (1, INF):
            return self._device_norms()[0 if p == INF else 1]
        E = self.E if E is None else E
        vg = np.maximum(0.0, np.maximum(E - pr.g_U, pr.g_L - E))
        vx = np.maximum(0.0, np.maximum(self.x - pr.x_U, pr.x_L - self.x))
        return np.linalg.norm(np.concatenate([vg, vx]), p)

    # slp.jl:54-66
    def compute_nu(self):
        if self.iter == 1:
            norm_df = 1.0 if self.feasibility_restoration else np.linalg.norm(self.df)
            self._ensure_optimizer()
            rn = self.optimizer.jac_row_norms()
            self.nu = np.maximum(1.0, norm_df / np.maximum(1.0, rn))
        else:
            self.nu = np.maximum(self.nu, np.abs(self.lam))

    def _slack_arrays(self):
        m = self.problem.m
        raw = getattr(self.p_slack, "raw", None)            # flat array of the C ABI (two entries per row, NaN = no second slack)
        if raw is not None and m:
            s2 = raw[1:2 * m:2]
            return raw[0:2 * m:2].copy(), np.where(np.isnan(s2), 0.0, s2)
        s1 = np.array([self.p_slack[i][0] for i in range(m)]) if m else np.zeros(0)
        s2 = np.array([self.p_slack[i][1] if len(self.p_slack[i]) > 1 else 0.0 for i in range(m)]) if m else np.zeros(0)
        return s1, s2

    # slp.jl:79-115
    def compute_phi(self, x, alpha, p):
        pr = self.problem
        if self._fm is not None and self._on_device(x):
            # the device evaluator holds the iterate of the last eval_functions(): the merit is taken there; any other x goes through the
            # host callbacks below (the reference's signature accepts any point)
            return self.optimizer.slp_merit(0, alpha, p, self.nu, self.p_slack, self.feasibility_restoration, self.prim_infeas)
        xp = x + alpha * p
        E = self.E if alpha == 0.0 else pr.eval_g(xp, np.zeros(pr.m))
        if self.feasibility_restoration:
            s1, s2 = self._slack_arrays()
            phi = self.prim_infeas + alpha * (s1.sum() + s2.sum())
            viol = np.maximum(0.0, np.maximum(self.E - pr.g_U, pr.g_L - self.E))
            lhs = E - viol
            lhs = lhs + alpha * np.where(self._both, s1 - s2, np.where(self._lo, s1, np.where(self._up, -s1, 0.0)))
            return float(phi + self.nu @ np.maximum(0.0, np.maximum(lhs - pr.g_U, pr.g_L - lhs)))
        phi = pr.eval_f(xp)
        return float(phi + self.nu @ np.maximum(0.0, np.maximum(E - pr.g_U, pr.g_L - E)))

    def _on_device(self, x):
        """x is the iterate of the last device evaluation (same object or an equal copy)."""
        xd = getattr(self, "_x_dev", None)
        return xd is not None and (x is xd or np.array_equal(x, xd))

    # slp.jl:122-147
    def compute_derivative(self):
        pr = self.problem
        if self._fm is not None:
            return self.optimizer.slp_merit(1, 0.0, self.p, self.nu, self.p_slack, self.feasibility_restoration, self.prim_infeas)
        viol = np.maximum(0.0, np.maximum(self.E - pr.g_U, pr.g_L - self.E))
        if self.feasibility_restoration:
            s1, s2 = self._slack_arrays()
            lhs = self.E - viol
            return float(s1.sum() + s2.sum() - self.nu @ np.maximum(0.0, np.maximum(lhs - pr.g_U, pr.g_L - lhs)))
        return float(self.df @ self.p - self.nu @ viol)

    def _start(self):
        pr = self.problem
        self.x[:] = pr.x
        lo = pr.x_L > -INF
        self.x[lo] = np.maximum(self.x[lo], pr.x_L[lo])
        hi = pr.x_U > -INF                                      # sic: slp_line_search.jl:102
        self.x[hi] = np.minimum(self.x[hi], pr.x_U[hi])

    def _finish(self):
        pr = self.problem
        pr.obj_val = pr.eval_f(self.x)
        pr.status = int(self.ret)
        pr.x[:] = self.x; pr.g[:] = self.E; pr.mult_g[:] = self.lam
        pr.mult_x_U[:] = self.mult_x_U; pr.mult_x_L[:] = self.mult_x_L
        pr.statistics.update(iter=self.iter, lp_solves=self.lp_solves, LP_time=self.lp_time)

    def _feasible_enough(self):
        return self.prim_infeas <= self.options.tol_infeas


class SlpLS(AbstractSlpOptimizer):
    """Sequential linear programming with line search (slp_line_search.jl)."""

    def __init__(self, problem):
        super().__init__(problem)
        self.alpha = 1.0
        self.directional_derivative = 0.0

    def compute_nu(self):                                        # slp_line_search.jl:251-261
        self.nu = np.abs(self.lam) if self.iter == 1 else np.maximum(self.nu, np.abs(self.lam))

    def compute_alpha(self):                                     # slp_line_search.jl:222-244
        o = self.options
        if self._fm is not None and self._on_device(self.x):
            # the trial points are evaluated on the device, eight per read-back (same alpha as the loop below)
            self.alpha, _, self.ls_trials, ok = self.optimizer.slp_line_search(self.p, self.nu, self.p_slack, self.feasibility_restoration, self.prim_infeas,
                                                                               self.phi, self.directional_derivative, o.eta, o.tau, o.min_alpha)
            if not ok and self.feasibility_restoration:
                self.ret = -3
            return ok
        self.alpha = 1.0
        while self.compute_phi(self.x, self.alpha, self.p) > self.phi + o.eta * self.alpha * self.directional_derivative:
            if self.alpha < o.min_alpha:
                if self.feasibility_restoration:
                    self.ret = -3
                return False
            self.alpha *= o.tau
        return True

    def run(self, max_lp_solves=None, resume=False):
        o = self.options
        if not resume:
            self._start()
            self.iter = 1
        while True:
            if max_lp_solves is not None and self.lp_solves >= max_lp_solves:
                break
            self.eval_functions()
            self.alpha = 0.0
            self.prim_infeas = self.norm_violations(INF)
            self.dual_infeas = self.KT_residuals()               # previous LP's multipliers (:117 before :122)
            self.compl = self.norm_complementarity()
            self.p, self.lam, self.mult_x_U, self.mult_x_L, self.p_slack, status = self.sub_optimize()
            if status not in (OPTIMAL, INFEASIBLE):
                if self._feasible_enough():
                    self.ret = 6
                break
            if status == INFEASIBLE:
                if self.feasibility_restoration:
                    self.ret = 6 if self._feasible_enough() else 2
                    break
                self.feasibility_restoration = True
                continue                                         # same x, iter not bumped (:145-146)
            self.compute_nu()
            self.phi = self.compute_phi(self.x, 0.0, self.p)
            self.directional_derivative = self.compute_derivative()
            is_valid_step = self.compute_alpha()
            if self.iter >= o.max_iter:
                self.ret = 6 if self._feasible_enough() else -1
                break
            if (self._feasible_enough() and self.compl <= o.tol_residual) or np.linalg.norm(self.p, INF) <= o.tol_direction:
                if self.feasibility_restoration:
                    self.feasibility_restoration = False
                    self.iter += 1
                    continue
                if self.dual_infeas <= o.tol_residual:
                    self.ret = 0
                    break
            if not is_valid_step:
                if self.ret == -3:
                    self.ret = 6 if self._feasible_enough() else 2
                    break
                self.feasibility_restoration = True
                self.iter += 1
                continue
            self.x = self.x + self.alpha * self.p
            self.iter += 1
        self._finish()


class SlpTR(AbstractSlpOptimizer):
    """Sequential linear programming with trust region (slp_trust_region.jl)."""

    def __init__(self, problem):
        super().__init__(problem)
        self.Delta = problem.parameters.tr_size                  # :62-65
        self.Delta_max, self.alpha1, self.alpha2 = 2.0, 0.1, 0.25

    def step_quality(self):                                      # :213-251
        o = self.options
        self.phi = self.compute_phi(self.x, 1.0, self.p) - self.compute_phi(self.x, 0.0, self.p)
        phi_pre = self.compute_derivative()
        if abs(phi_pre) > 0.0:
            rho = self.phi / phi_pre
            if rho <= 0:
                self.Delta *= self.alpha1
            elif rho <= 0.25:
                self.Delta *= self.alpha2
            elif rho > 0.75:
                self.Delta = min(2 * self.Delta, self.Delta_max)
            return rho
        rho = -self.phi
        if abs(self.phi) < 1.e-8:
            if self.feasibility_restoration:
                self.feasibility_restoration = False
            elif self._feasible_enough():
                self.ret = 0 if (self.dual_infeas <= o.tol_residual and self.compl <= o.tol_residual) else 6
            else:
                self.ret = 2
        return rho

    def run(self, max_lp_solves=None, resume=False):
        o, pr = self.options, self.problem
        if not resume:
            self._start()
            self.iter = 1
        while True:
            if max_lp_solves is not None and self.lp_solves >= max_lp_solves:
                break
            self.eval_functions()
            self.p, self.lam, self.mult_x_U, self.mult_x_L, self.p_slack, status = self.sub_optimize(self.Delta)
            if status not in (OPTIMAL, INFEASIBLE):
                if self.norm_violations(1, pr.eval_g(self.x, np.zeros(pr.m))) <= o.tol_infeas:
                    self.ret = 6
                break
            if status == INFEASIBLE:
                if self.feasibility_restoration:
                    self.ret = 6 if self._feasible_enough() else 2
                    break
                self.feasibility_restoration = True
                continue
            self.compute_nu()
            self.prim_infeas = self.norm_violations(INF)
            self.dual_infeas = self.KT_residuals()
            self.compl = self.norm_complementarity()
            if self._feasible_enough() and self.compl <= o.tol_residual and np.linalg.norm(self.p, INF) <= o.tol_direction:
                if self.feasibility_restoration:
                    self.feasibility_restoration = False
                    self.iter += 1
                    continue
                if self.dual_infeas <= o.tol_residual:
                    self.ret = 0
                    break
            if self.iter >= o.max_iter:
                self.ret = 6 if self._feasible_enough() else -1
                break
            rho = self.step_quality()
            if self.ret in (0, 2, 6):
                break
            if rho >= 0:
                self.x = self.x + self.p
            self.iter += 1
        self._finish()


def optimize(model, max_lp_solves=None):
    """src/model.jl:63-80 (an unset `external_optimizer` selects the HIP sub-optimizer instead of status -12)."""
    par = model.parameters
    if par.method != "SLP":
        raise ValueError("The method is not defined")
    if par.algorithm == "Line Search":
        slp = SlpLS(model)
    elif par.algorithm == "Trust Region":
        slp = SlpTR(model)
    else:
        raise ValueError("unknown algorithm %r" % par.algorithm)
    slp.run(max_lp_solves)
    return slp
