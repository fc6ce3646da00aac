"""Oracle restatement of the SLP callers and their shared numerics
(TEST INFRASTRUCTURE - see oracle/__init__.py).

  Parameters                      src/parameters.jl:1-29 (defaults are contract)
  Model                           src/model.jl:1-61
  KT_residuals / norms            src/algorithms/common.jl:35-98
  LpData, sub_optimize!(slp, Δ)   src/algorithms/slp.jl:8-47
  compute_nu!, compute_phi, compute_derivative, eval_functions!   src/algorithms/slp.jl:54-193
  SlpLS.run!, compute_alpha       src/algorithms/slp_line_search.jl:78-261
  SlpTR.run!, step_quality        src/algorithms/slp_trust_region.jl:87-251
Status codes: src/status.jl:2-22.
"""
import numpy as np
from . import lp_solver as L
from .subproblem import QpData, QpModel, compute_jacobian_matrix

INF = np.inf


class Parameters:
    """src/parameters.jl:1-29."""

    def __init__(self, **kw):
        self.method = "SLP"
        self.algorithm = "Line Search"
        self.OutputFlag = 0
        self.StatisticsFlag = 0
        self.tol_direction = 1.e-6
        self.tol_residual = 0.01
        self.tol_infeas = 0.01
        self.max_iter = 1000
        self.eta = 0.4
        self.tau = 0.9
        self.min_alpha = 1.e-6
        self.tr_size = 0.4
        for k, v in kw.items():
            if not hasattr(self, k):
                raise KeyError(k)
            setattr(self, k, v)


class Model:
    """src/model.jl:1-61.  Callbacks follow the reference's signatures:
    eval_f(x)->f ; eval_grad_f(x, out) ; eval_g(x, out)->out ; eval_jac_g(x, out_values)."""

    def __init__(self, n, m, x_L, x_U, g_L, g_U, j_str, eval_f, eval_g, eval_grad_f, eval_jac_g, parameters=None):
        self.n, self.m = n, m
        self.x = np.zeros(n)
        self.x_L, self.x_U = np.asarray(x_L, float), np.asarray(x_U, float)
        self.g = np.zeros(m)
        self.g_L, self.g_U = np.asarray(g_L, float), np.asarray(g_U, float)
        self.j_str = j_str                              # list of 1-based (row, col)
        self.mult_g = np.zeros(m); self.mult_x_L = np.zeros(n); self.mult_x_U = np.zeros(n)
        self.obj_val = 0.0
        self.status = -5
        self.eval_f, self.eval_g, self.eval_grad_f, self.eval_jac_g = eval_f, eval_g, eval_grad_f, eval_jac_g
        self.parameters = parameters or Parameters()
        self.statistics = {}


# ------------------------------------------------------------------ common.jl
def KT_residuals(df, lam, mult_x_U, mult_x_L, Jac):
    """common.jl:35-44."""
    KT_res = np.linalg.norm(df - Jac.T @ lam - mult_x_U - mult_x_L)
    scalar = max(1.0, np.linalg.norm(df))
    for i in range(Jac.shape[0]):
        scalar = max(scalar, abs(lam[i]) * np.linalg.norm(Jac[i, :]))
    return KT_res / scalar


def norm_complementarity(E, g_L, g_U, lam, p=INF):
    """common.jl:51-68."""
    m = len(E)
    compl = np.zeros(m)
    denom = 0.0
    for i in range(m):
        if g_L[i] == g_U[i]:
            compl[i] = 0.0
        else:
            compl[i] = min(E[i] - g_L[i], g_U[i] - E[i]) * lam[i]
            denom += lam[i] ** 2
    nrm = np.linalg.norm(compl, p) if m > 0 else 0.0
    return nrm / (1 + np.sqrt(denom))


def norm_violations(E, g_L, g_U, x, x_L, x_U, p=1):
    """common.jl:75-98."""
    m, n = len(E), len(x)
    viol = np.zeros(m + n)
    for i in range(m):
        if E[i] > g_U[i]:
            viol[i] = E[i] - g_U[i]
        elif E[i] < g_L[i]:
            viol[i] = g_L[i] - E[i]
    for j in range(n):
        if x[j] > x_U[j]:
            viol[m + j] = x[j] - x_U[j]
        elif x[j] < x_L[j]:
            viol[m + j] = x_L[j] - x[j]
    return np.linalg.norm(viol, p)


# ------------------------------------------------------------------ slp.jl shared pieces
class _Slp:
    def __init__(self, problem):
        n, m = problem.n, problem.m
        self.problem = problem
        self.x = np.zeros(n); self.p = np.zeros(n); self.p_slack = {}
        self.lam = np.zeros(m); self.mult_x_L = np.zeros(n); self.mult_x_U = np.zeros(n)
        self.f = 0.0; self.df = np.zeros(n); self.E = np.zeros(m); self.dE = np.zeros(len(problem.j_str))
        self.phi = INF; self.nu = np.zeros(m)
        self.prim_infeas = INF; self.dual_infeas = INF; self.compl = INF
        self.options = problem.parameters
        self.optimizer = None
        self.feasibility_restoration = False
        self.iter = 1; self.ret = -5
        self.j_row = np.array([rc[0] for rc in problem.j_str], np.int64)
        self.j_col = np.array([rc[1] for rc in problem.j_str], np.int64)
        self.trace = []          # per LP solve: dict(iter, fr, status, active sets, p, lam, ...)
        self.lp_solves = 0

    # slp.jl:186-193
    def eval_functions(self):
        pr = self.problem
        self.f = pr.eval_f(self.x)
        pr.eval_grad_f(self.x, self.df)
        pr.eval_g(self.x, self.E)
        pr.eval_jac_g(self.x, self.dE)

    def jacobian(self):
        pr = self.problem
        return compute_jacobian_matrix(pr.m, pr.n, self.j_row - 1, self.j_col - 1, self.dE)

    # slp.jl:8-47
    def sub_optimize(self, Delta=1000.0):
        pr = self.problem
        A, stored = self.jacobian()
        data = QpData(self.df.copy(), self.f, A, self.E.copy(), pr.g_L, pr.g_U, pr.x_L, pr.x_U, stored)
        if self.optimizer is None:
            self.optimizer = QpModel(data, self.j_row, self.j_col)
        else:
            self.optimizer.data = data
        out = self.optimizer.sub_optimize(self.x, Delta, self.feasibility_restoration)
        self.lp_solves += 1
        Xsol, lam, mU, mL, p_slack, status, info = out
        rec = dict(iter=self.iter, fr=bool(self.feasibility_restoration), status=int(status), delta=float(Delta),
                   p=Xsol.copy(), lam=lam.copy(), mult_x_U=mU.copy(), mult_x_L=mL.copy(), x=self.x.copy(),
                   stats=dict(info['stats']))
        if status == L.OPTIMAL:
            rec['sets'] = tuple(a.copy() for a in info['sets'])
        self.trace.append(rec)
        return Xsol, lam, mU, mL, p_slack, status

    # slp.jl:54-66 (used by SlpTR; SlpLS overrides)
    def compute_nu(self):
        pr = self.problem
        if self.iter == 1:
            norm_df = 1.0 if self.feasibility_restoration else np.linalg.norm(self.df)
            J, _ = self.jacobian()
            for i in range(pr.m):
                self.nu[i] = max(1.0, norm_df / max(1.0, np.linalg.norm(J[i, :])))
        else:
            for i in range(pr.m):
                self.nu[i] = max(self.nu[i], abs(self.lam[i]))

    # slp.jl:79-115
    def compute_phi(self, x, alpha, p):
        pr = self.problem
        xp = x + alpha * p
        E = self.E if alpha == 0.0 else pr.eval_g(xp, np.zeros(pr.m))
        if self.feasibility_restoration:
            phi = self.prim_infeas
            for i, v in self.p_slack.items():
                phi += alpha * sum(v)
            for i in range(pr.m):
                viol = max(0.0, self.E[i] - pr.g_U[i], pr.g_L[i] - self.E[i])
                lhs = E[i] - viol
                if pr.g_L[i] > -INF and pr.g_U[i] < INF:
                    lhs += alpha * (self.p_slack[i][0] - self.p_slack[i][1])
                elif pr.g_L[i] > -INF:
                    lhs += alpha * self.p_slack[i][0]
                elif pr.g_U[i] < INF:
                    lhs -= alpha * self.p_slack[i][0]
                phi += self.nu[i] * max(0.0, lhs - pr.g_U[i], pr.g_L[i] - lhs)
        else:
            phi = pr.eval_f(xp)
            for i in range(pr.m):
                if E[i] > pr.g_U[i]:
                    phi += self.nu[i] * (E[i] - pr.g_U[i])
                elif E[i] < pr.g_L[i]:
                    phi += self.nu[i] * (pr.g_L[i] - E[i])
        return phi

    # slp.jl:122-147
    def compute_derivative(self):
        pr = self.problem
        D = 0.0
        if self.feasibility_restoration:
            for i, v in self.p_slack.items():
                D += sum(v)
            for i in range(pr.m):
                viol = max(0.0, self.E[i] - pr.g_U[i], pr.g_L[i] - self.E[i])
                lhs = self.E[i] - viol
                D -= self.nu[i] * max(0.0, lhs - pr.g_U[i], pr.g_L[i] - lhs)
        else:
            D = float(self.df @ self.p)
            for i in range(pr.m):
                if self.E[i] > pr.g_U[i]:
                    D -= self.nu[i] * (self.E[i] - pr.g_U[i])
                elif self.E[i] < pr.g_L[i]:
                    D -= self.nu[i] * (pr.g_L[i] - self.E[i])
        return D

    def KT_residuals(self):
        J, _ = self.jacobian()
        return KT_residuals(self.df, self.lam, self.mult_x_U, self.mult_x_L, J)

    def norm_complementarity(self, p=INF):
        pr = self.problem
        return norm_complementarity(self.E, pr.g_L, pr.g_U, self.lam, p)

    def norm_violations(self, p=1):
        pr = self.problem
        return norm_violations(self.E, pr.g_L, pr.g_U, self.x, pr.x_L, pr.x_U, p)

    def _init_x(self):
        pr = self.problem
        self.x[:] = pr.x
        for i in range(pr.n):                      # slp_line_search.jl:98-105 (note `x_U > -Inf`)
            if pr.x_L[i] > -INF:
                self.x[i] = max(self.x[i], pr.x_L[i])
            if pr.x_U[i] > -INF:
                self.x[i] = min(self.x[i], pr.x_U[i])

    def _epilogue(self):
        pr = self.problem
        pr.obj_val = pr.eval_f(self.x)
        pr.status = int(self.ret)
        pr.x[:] = self.x
        pr.g[:] = self.E
        pr.mult_g[:] = self.lam
        pr.mult_x_U[:] = self.mult_x_U
        pr.mult_x_L[:] = self.mult_x_L
        pr.statistics['iter'] = self.iter
        pr.statistics['lp_solves'] = self.lp_solves


class SlpLS(_Slp):
    """slp_line_search.jl:4-261."""

    def __init__(self, problem):
        super().__init__(problem)
        self.alpha = 1.0
        self.directional_derivative = 0.0

    def compute_nu(self):                          # slp_line_search.jl:251-261
        if self.iter == 1:
            self.nu[:] = np.abs(self.lam)
        else:
            self.nu[:] = np.maximum(self.nu, np.abs(self.lam))

    def compute_alpha(self):                       # slp_line_search.jl:222-244
        o = self.options
        is_valid = True
        self.alpha = 1.0
        phi_x_p = self.compute_phi(self.x, self.alpha, self.p)
        while phi_x_p > self.phi + o.eta * self.alpha * self.directional_derivative:
            if self.alpha < o.min_alpha:
                if self.feasibility_restoration:
                    self.ret = -3
                is_valid = False
                break
            self.alpha *= o.tau
            phi_x_p = self.compute_phi(self.x, self.alpha, self.p)
        return is_valid

    def run(self):                                 # slp_line_search.jl:78-215
        o = self.options
        self._init_x()
        self.iter = 1
        while True:
            self.eval_functions()
            self.alpha = 0.0
            self.prim_infeas = self.norm_violations(INF)
            self.dual_infeas = self.KT_residuals()
            self.compl = self.norm_complementarity()
            self.p, self.lam, self.mult_x_U, self.mult_x_L, self.p_slack, status = self.sub_optimize()
            if status not in (L.OPTIMAL, L.INFEASIBLE):
                # `slp.ret == -3` is a comparison in the reference (:129)
                if self.prim_infeas <= o.tol_infeas:
                    self.ret = 6
                break
            elif status == L.INFEASIBLE:
                if self.feasibility_restoration:
                    self.ret = 6 if self.prim_infeas <= o.tol_infeas else 2
                    break
                else:
                    self.feasibility_restoration = True
                    continue
            self.compute_nu()
            self.phi = self.compute_phi(self.x, 0.0, self.p)
            self.directional_derivative = self.compute_derivative()
            is_valid_step = self.compute_alpha()
            if self.iter >= o.max_iter:
                self.ret = -1
                if self.prim_infeas <= o.tol_infeas:
                    self.ret = 6
                break
            if (self.prim_infeas <= o.tol_infeas and self.compl <= o.tol_residual) or \
                    np.linalg.norm(self.p, INF) <= o.tol_direction:
                if self.feasibility_restoration:
                    self.feasibility_restoration = False
                    self.iter += 1
                    continue
                elif self.dual_infeas <= o.tol_residual:
                    self.ret = 0
                    break
            if not is_valid_step:
                if self.ret == -3:
                    self.ret = 6 if self.prim_infeas <= o.tol_infeas else 2
                    break
                else:
                    self.feasibility_restoration = True
                self.iter += 1
                continue
            self.x = self.x + self.alpha * self.p
            self.iter += 1
        self._epilogue()


class SlpTR(_Slp):
    """slp_trust_region.jl:10-251."""

    def __init__(self, problem):
        super().__init__(problem)
        self.Delta = problem.parameters.tr_size
        self.Delta_max = 2.0
        self.alpha1 = 0.1
        self.alpha2 = 0.25

    def step_quality(self):                        # slp_trust_region.jl:213-251
        o = self.options
        self.phi = self.compute_phi(self.x, 1.0, self.p) - self.compute_phi(self.x, 0.0, self.p)
        phi_pre = self.compute_derivative()
        rho = 0.0
        if abs(phi_pre) > 0.0:
            rho = self.phi / phi_pre
            if rho <= 0:
                self.Delta *= self.alpha1
            elif rho <= 0.25:
                self.Delta *= self.alpha2
            elif rho > 0.75:
                self.Delta = min(2 * self.Delta, self.Delta_max)
        else:
            rho = -self.phi
            if abs(self.phi) < 1.e-8:
                if self.feasibility_restoration:
                    self.feasibility_restoration = False
                else:
                    if self.prim_infeas <= o.tol_infeas:
                        if self.dual_infeas <= o.tol_residual and self.compl <= o.tol_residual:
                            self.ret = 0
                        else:
                            self.ret = 6
                    else:
                        self.ret = 2
        return rho

    def run(self):                                 # slp_trust_region.jl:87-206
        o = self.options
        pr = self.problem
        self._init_x()
        self.iter = 1
        while True:
            self.eval_functions()
            self.p, self.lam, self.mult_x_U, self.mult_x_L, self.p_slack, status = self.sub_optimize(self.Delta)
            if status not in (L.OPTIMAL, L.INFEASIBLE):
                Ex = pr.eval_g(self.x, np.zeros(pr.m))
                if norm_violations(Ex, pr.g_L, pr.g_U, self.x, pr.x_L, pr.x_U, 1) <= o.tol_infeas:
                    self.ret = 6
                break
            elif status == L.INFEASIBLE:
                if self.feasibility_restoration:
                    self.ret = 6 if self.prim_infeas <= o.tol_infeas else 2
                    break
                else:
                    self.feasibility_restoration = True
                    continue
            self.compute_nu()
            self.prim_infeas = self.norm_violations(INF)
            self.dual_infeas = self.KT_residuals()
            self.compl = self.norm_complementarity()
            if self.prim_infeas <= o.tol_infeas and self.compl <= o.tol_residual and \
                    np.linalg.norm(self.p, INF) <= o.tol_direction:
                if self.feasibility_restoration:
                    self.feasibility_restoration = False
                    self.iter += 1
                    continue
                elif self.dual_infeas <= o.tol_residual:
                    self.ret = 0
                    break
            if self.iter >= o.max_iter:
                self.ret = -1
                if self.prim_infeas <= o.tol_infeas:
                    self.ret = 6
                break
            rho = self.step_quality()
            if self.ret in (0, 2, 6):
                break
            if rho >= 0:
                self.x = self.x + self.p
            self.iter += 1
        self._epilogue()


def optimize(model):
    """src/model.jl:63-80."""
    if model.parameters.method != "SLP":
        raise ValueError("The method is not defined")
    slp = SlpLS(model) if model.parameters.algorithm == "Line Search" else SlpTR(model)
    slp.run()
    return slp
