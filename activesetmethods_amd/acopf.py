"""Polar AC optimal power flow as the reference's ACOPF driver poses it (inputs to the SLP hot path).

The reference builds this model with PowerModels' `ACPPowerModel` + `build_opf` (test/opf.jl:6-10,
examples/acopf/opf.jl:18-36) and reaches the hot path through the MOI wrapper's evaluator
(src/MOI_wrapper.jl:683-944).  PowerModels / JuMP are not available offline, so the formulation is
restated here directly:

  variables   va, vm (per bus) | pg, qg (per generator) | p, q (per arc: from-side then to-side) |
              p_dc, q_dc (per dc-line arc)                 -> n = 2 n_bus + 2 n_gen + 4 n_branch + 4 n_dc
  rows, in the wrapper's block order linear <=, >=, ==, quadratic <=, ==, NLP (MOI_wrapper.jl:683-689):
    angle-difference upper / lower (linear), reference angle, dc-line loss (linear ==),
    thermal limits p^2 + q^2 <= rate^2 (quadratic <=, both ends), nodal P and Q balance (== ; the shunt
    term g_s vm^2 makes it quadratic), Ohm's law p_fr, q_fr, p_to, q_to (NLP ==, 4 per branch)
  objective   sum of polynomial generation costs (per-unit scaled)

The row order *inside* a block and the variable order follow PowerModels' creation order where it is
documented and are otherwise ours: they only permute rows/columns of the LP.

`case_from_tables` takes MATPOWER-format tables (bus, gen, gencost, branch, dcline, baseMVA);
`synthetic_grid` makes seeded stand-ins with the bus/gen/branch counts of the public cases that are not
shipped with the reference (case118, case300, case1354pegase) - SURVEY.md section 8d.
"""
import numpy as np

from .problems import Problem, splitmix64, uniform01

INF = np.inf


def case_from_tables(baseMVA, bus, gen, gencost, branch, dcline=None):
    """MATPOWER tables -> per-unit dict (in-service elements only)."""
    bus = np.atleast_2d(np.asarray(bus, float))
    gen = np.atleast_2d(np.asarray(gen, float))
    gencost = np.atleast_2d(np.asarray(gencost, float))
    branch = np.atleast_2d(np.asarray(branch, float))
    dcline = np.zeros((0, 17)) if dcline is None or len(dcline) == 0 else np.atleast_2d(np.asarray(dcline, float))
    ids = bus[:, 0].astype(int)
    pos = {b: i for i, b in enumerate(ids)}
    on = gen[:, 7] > 0
    gen, gencost = gen[on], gencost[on]
    branch = branch[branch[:, 10] > 0]
    dcline = dcline[dcline[:, 2] > 0] if len(dcline) else dcline
    c = dict(baseMVA=float(baseMVA), n_bus=len(bus))
    c["bus_type"] = bus[:, 1].astype(int)
    c["pd"] = bus[:, 2] / baseMVA; c["qd"] = bus[:, 3] / baseMVA
    c["gs"] = bus[:, 4] / baseMVA; c["bs"] = bus[:, 5] / baseMVA
    c["vmax"] = bus[:, 11]; c["vmin"] = bus[:, 12]
    c["gen_bus"] = np.array([pos[int(b)] for b in gen[:, 0]], int)
    c["qmax"] = gen[:, 3] / baseMVA; c["qmin"] = gen[:, 4] / baseMVA
    c["pmax"] = gen[:, 8] / baseMVA; c["pmin"] = gen[:, 9] / baseMVA
    # polynomial cost (model 2): n coefficients, highest order first; keep up to quadratic, per-unit scaled
    ncost = gencost[:, 3].astype(int)
    c2 = np.zeros(len(gen)); c1 = np.zeros(len(gen)); c0 = np.zeros(len(gen))
    for g in range(len(gen)):
        co = gencost[g, 4:4 + ncost[g]]
        co = np.concatenate([np.zeros(3 - len(co)), co])[-3:] if len(co) <= 3 else co[-3:]
        c2[g], c1[g], c0[g] = co[0] * baseMVA ** 2, co[1] * baseMVA, co[2]
    c["cost2"], c["cost1"], c["cost0"] = c2, c1, c0
    c["f_bus"] = np.array([pos[int(b)] for b in branch[:, 0]], int)
    c["t_bus"] = np.array([pos[int(b)] for b in branch[:, 1]], int)
    r, x, bc = branch[:, 2], branch[:, 3], branch[:, 4]
    y = 1.0 / (r + 1j * x)
    c["g"], c["b"] = y.real, y.imag
    c["b_fr"] = bc / 2.0; c["b_to"] = bc / 2.0; c["g_fr"] = np.zeros(len(branch)); c["g_to"] = np.zeros(len(branch))
    tap = np.where(branch[:, 8] == 0, 1.0, branch[:, 8])
    shift = np.deg2rad(branch[:, 9])
    c["tr"], c["ti"], c["tm"] = tap * np.cos(shift), tap * np.sin(shift), tap
    rate = branch[:, 5] / baseMVA
    c["rate_a"] = np.where(rate > 0, rate, INF)
    c["angmin"] = np.deg2rad(branch[:, 11]); c["angmax"] = np.deg2rad(branch[:, 12])
    nd = len(dcline)
    c["dc_f"] = np.array([pos[int(b)] for b in dcline[:, 0]], int) if nd else np.zeros(0, int)
    c["dc_t"] = np.array([pos[int(b)] for b in dcline[:, 1]], int) if nd else np.zeros(0, int)
    if nd:
        pmin, pmax = dcline[:, 9] / baseMVA, dcline[:, 10] / baseMVA
        loss0, loss1 = dcline[:, 15] / baseMVA, dcline[:, 16]
        # MATPOWER limits are on the from-side flow; to-side follows from the loss model (pmin, pmax >= 0 case)
        c["dc_pminf"], c["dc_pmaxf"] = pmin, pmax
        c["dc_pmint"], c["dc_pmaxt"] = loss0 - pmax * (1 - loss1), loss0 - pmin * (1 - loss1)
        c["dc_qminf"], c["dc_qmaxf"] = dcline[:, 11] / baseMVA, dcline[:, 12] / baseMVA
        c["dc_qmint"], c["dc_qmaxt"] = dcline[:, 13] / baseMVA, dcline[:, 14] / baseMVA
        c["dc_loss0"], c["dc_loss1"] = loss0, loss1
    ref = np.nonzero(c["bus_type"] == 3)[0]
    if len(ref) == 0:                       # no reference bus in the file: bus of the largest generator (first on ties)
        ref = np.array([c["gen_bus"][int(np.argmax(c["pmax"]))]])
    c["ref"] = ref
    return c


def synthetic_grid(n_bus, n_gen, n_branch, seed=1, load_scale=1.0):
    """Seeded stand-in grid: a ring plus random chords; r, x, b, loads and generator data from stated
    ranges.  NOT a real network - only the element counts match the public cases."""
    u = uniform01(seed, 16 * (n_bus + n_gen + n_branch) + 64)
    z = splitmix64(seed + 7, 4 * n_branch + n_gen + 16)
    k = [0]

    def U(cnt, lo, hi):
        out = lo + (hi - lo) * u[k[0]:k[0] + cnt]
        k[0] += cnt
        return out
    f = np.arange(n_bus)
    t = (f + 1) % n_bus
    extra = n_branch - n_bus
    if extra > 0:
        ef = (z[:extra] % np.uint64(n_bus)).astype(int)
        span = 2 + (z[extra:2 * extra] % np.uint64(max(n_bus // 8, 3))).astype(int)
        et = (ef + span) % n_bus
        f = np.concatenate([f, ef]); t = np.concatenate([t, et])
    else:
        f, t = f[:n_branch], t[:n_branch]
    nl = len(f)
    r = U(nl, 0.002, 0.03); x = U(nl, 0.02, 0.25); bc = U(nl, 0.0, 0.08)
    pd = U(n_bus, 0.0, 0.9) * load_scale
    qd = pd * U(n_bus, 0.1, 0.4)
    gen_bus = np.sort((z[2 * max(extra, 0) + 1: 2 * max(extra, 0) + 1 + n_gen] % np.uint64(n_bus)).astype(int))
    total = pd.sum()
    cap = U(n_gen, 0.5, 1.5)
    pmax = cap / cap.sum() * total * 2.5
    bus = np.zeros((n_bus, 13))
    bus[:, 0] = np.arange(1, n_bus + 1); bus[:, 1] = 1
    bus[:, 2] = pd * 100; bus[:, 3] = qd * 100; bus[:, 7] = 1.0; bus[:, 11] = 1.1; bus[:, 12] = 0.9
    bus[gen_bus[int(np.argmax(pmax))], 1] = 3
    gen = np.zeros((n_gen, 10))
    gen[:, 0] = gen_bus + 1; gen[:, 3] = 0.6 * pmax * 100; gen[:, 4] = -0.6 * pmax * 100
    gen[:, 5] = 1.0; gen[:, 6] = 100; gen[:, 7] = 1; gen[:, 8] = pmax * 100; gen[:, 9] = 0.0
    gencost = np.zeros((n_gen, 7))
    gencost[:, 0] = 2; gencost[:, 3] = 3
    gencost[:, 4] = U(n_gen, 0.005, 0.12); gencost[:, 5] = U(n_gen, 1.0, 30.0)
    branch = np.zeros((nl, 13))
    branch[:, 0] = f + 1; branch[:, 1] = t + 1; branch[:, 2] = r; branch[:, 3] = x; branch[:, 4] = bc
    branch[:, 5] = U(nl, 2.0, 6.0) * 100 * max(1.0, total / n_bus)
    branch[:, 10] = 1; branch[:, 11] = -30; branch[:, 12] = 30
    return case_from_tables(100.0, bus, gen, gencost, branch)


class AcopfModel:
    """Index maps + vectorised evaluators of the polar ACOPF for one case dict."""

    def __init__(self, c):
        self.c = c
        nb, ng, nl, nd = c["n_bus"], len(c["gen_bus"]), len(c["f_bus"]), len(c["dc_f"])
        self.nb, self.ng, self.nl, self.nd = nb, ng, nl, nd
        o = 0
        self.va = np.arange(o, o + nb); o += nb
        self.vm = np.arange(o, o + nb); o += nb
        self.pg = np.arange(o, o + ng); o += ng
        self.qg = np.arange(o, o + ng); o += ng
        self.pf = np.arange(o, o + nl); o += nl            # p on from-side arcs
        self.pt = np.arange(o, o + nl); o += nl            # p on to-side arcs
        self.qf = np.arange(o, o + nl); o += nl
        self.qt = np.arange(o, o + nl); o += nl
        self.pdf = np.arange(o, o + nd); o += nd
        self.pdt = np.arange(o, o + nd); o += nd
        self.qdf = np.arange(o, o + nd); o += nd
        self.qdt = np.arange(o, o + nd); o += nd
        self.n = o
        # ---- bounds and start (examples/acopf/init_opf.jl:25-29: midpoint of the bounds where both exist)
        xl = np.full(o, -INF); xu = np.full(o, INF)
        xl[self.vm], xu[self.vm] = c["vmin"], c["vmax"]
        xl[self.pg], xu[self.pg] = c["pmin"], c["pmax"]
        xl[self.qg], xu[self.qg] = c["qmin"], c["qmax"]
        for idx in (self.pf, self.pt, self.qf, self.qt):
            xl[idx], xu[idx] = -c["rate_a"], c["rate_a"]
        if nd:
            xl[self.pdf], xu[self.pdf] = c["dc_pminf"], c["dc_pmaxf"]
            xl[self.pdt], xu[self.pdt] = c["dc_pmint"], c["dc_pmaxt"]
            xl[self.qdf], xu[self.qdf] = c["dc_qminf"], c["dc_qmaxf"]
            xl[self.qdt], xu[self.qdt] = c["dc_qmint"], c["dc_qmaxt"]
        self.x_L, self.x_U = xl, xu
        both = np.isfinite(xl) & np.isfinite(xu)
        x0 = np.zeros(o)
        x0[self.vm] = 1.0
        x0[both] = 0.5 * (xl[both] + xu[both])
        self.x0 = x0
        # ---- rows
        f, t = c["f_bus"], c["t_bus"]
        lim = np.isfinite(c["rate_a"])
        self.lim = np.nonzero(lim)[0]
        nlim = len(self.lim)
        nref = len(c["ref"])
        r = 0
        self.r_angu = np.arange(r, r + nl); r += nl          # va_f - va_t <= angmax
        self.r_angl = np.arange(r, r + nl); r += nl          # va_f - va_t >= angmin
        self.r_ref = np.arange(r, r + nref); r += nref       # va_ref == 0
        self.r_dc = np.arange(r, r + nd); r += nd            # (1-loss1) p_f + p_t == loss0
        self.r_thf = np.arange(r, r + nlim); r += nlim       # p_f^2 + q_f^2 <= rate^2
        self.r_tht = np.arange(r, r + nlim); r += nlim
        self.r_pb = np.arange(r, r + nb); r += nb            # nodal P balance
        self.r_qb = np.arange(r, r + nb); r += nb            # nodal Q balance
        self.r_pfr = np.arange(r, r + nl); r += nl           # Ohm's law (NLP block)
        self.r_qfr = np.arange(r, r + nl); r += nl
        self.r_pto = np.arange(r, r + nl); r += nl
        self.r_qto = np.arange(r, r + nl); r += nl
        self.m = r
        gl = np.zeros(r); gu = np.zeros(r)
        gl[self.r_angu] = -INF; gu[self.r_angu] = c["angmax"]
        gl[self.r_angl] = c["angmin"]; gu[self.r_angl] = INF
        if nd:
            gl[self.r_dc] = gu[self.r_dc] = c["dc_loss0"]
        gl[self.r_thf] = -INF; gu[self.r_thf] = c["rate_a"][self.lim] ** 2
        gl[self.r_tht] = -INF; gu[self.r_tht] = c["rate_a"][self.lim] ** 2
        gl[self.r_pb] = gu[self.r_pb] = c["pd"]              # sum(pg) - sum(p arcs) - gs vm^2 == pd
        gl[self.r_qb] = gu[self.r_qb] = c["qd"]
        self.g_L, self.g_U = gl, gu
        # ---- Jacobian pattern (row, col) in evaluation order; values are produced in the same order
        R, C = [], []

        def add(rows, cols):
            R.append(np.asarray(rows)); C.append(np.asarray(cols))
        add(self.r_angu, self.va[f]); add(self.r_angu, self.va[t])
        add(self.r_angl, self.va[f]); add(self.r_angl, self.va[t])
        add(self.r_ref, self.va[c["ref"]])
        if nd:
            add(self.r_dc, self.pdf); add(self.r_dc, self.pdt)
        L = self.lim
        add(self.r_thf, self.pf[L]); add(self.r_thf, self.qf[L])
        add(self.r_tht, self.pt[L]); add(self.r_tht, self.qt[L])
        gb = c["gen_bus"]
        add(self.r_pb[gb], self.pg); add(self.r_pb[f], self.pf); add(self.r_pb[t], self.pt)
        add(self.r_qb[gb], self.qg); add(self.r_qb[f], self.qf); add(self.r_qb[t], self.qt)
        if nd:
            add(self.r_pb[c["dc_f"]], self.pdf); add(self.r_pb[c["dc_t"]], self.pdt)
            add(self.r_qb[c["dc_f"]], self.qdf); add(self.r_qb[c["dc_t"]], self.qdt)
        self.sh = np.nonzero((c["gs"] != 0) | (c["bs"] != 0))[0]
        add(self.r_pb[self.sh], self.vm[self.sh]); add(self.r_qb[self.sh], self.vm[self.sh])
        for rr, pv in ((self.r_pfr, self.pf), (self.r_qfr, self.qf), (self.r_pto, self.pt), (self.r_qto, self.qt)):
            add(rr, pv); add(rr, self.vm[f]); add(rr, self.vm[t]); add(rr, self.va[f]); add(rr, self.va[t])
        self.j_row = np.concatenate(R).astype(np.int64) + 1
        self.j_col = np.concatenate(C).astype(np.int64) + 1
        # branch admittance combinations
        g, b, tr, ti, tm = c["g"], c["b"], c["tr"], c["ti"], c["tm"]
        tm2 = tm * tm
        self.k_ff_p = (g + c["g_fr"]) / tm2; self.k_ff_q = -(b + c["b_fr"]) / tm2
        self.k_tt_p = (g + c["g_to"]); self.k_tt_q = -(b + c["b_to"])
        self.a_f = (-g * tr + b * ti) / tm2; self.b_f = (-b * tr - g * ti) / tm2      # from side: cos / sin coefficients
        self.a_t = (-g * tr - b * ti) / tm2; self.b_t = (-b * tr + g * ti) / tm2      # to side

    # ---- flows (Ohm's law right-hand sides)
    def _flows(self, x):
        c = self.c
        f, t = c["f_bus"], c["t_bus"]
        vf, vt = x[self.vm[f]], x[self.vm[t]]
        d = x[self.va[f]] - x[self.va[t]]
        cs, sn = np.cos(d), np.sin(d)
        vv = vf * vt
        pfr = self.k_ff_p * vf * vf + self.a_f * vv * cs + self.b_f * vv * sn
        qfr = self.k_ff_q * vf * vf - self.b_f * vv * cs + self.a_f * vv * sn
        pto = self.k_tt_p * vt * vt + self.a_t * vv * cs - self.b_t * vv * sn          # cos(-d)=cs, sin(-d)=-sn
        qto = self.k_tt_q * vt * vt - self.b_t * vv * cs - self.a_t * vv * sn
        return vf, vt, cs, sn, vv, pfr, qfr, pto, qto

    def eval_f(self, x):
        c = self.c
        pg = x[self.pg]
        return float(np.sum(c["cost2"] * pg * pg + c["cost1"] * pg + c["cost0"]))

    def eval_grad_f(self, x, g):
        c = self.c
        g[:] = 0.0
        g[self.pg] = 2.0 * c["cost2"] * x[self.pg] + c["cost1"]
        return g

    def eval_g(self, x, out):
        c = self.c
        f, t, nd = c["f_bus"], c["t_bus"], self.nd
        d = x[self.va[f]] - x[self.va[t]]
        out[self.r_angu] = d
        out[self.r_angl] = d
        out[self.r_ref] = x[self.va[c["ref"]]]
        if nd:
            out[self.r_dc] = (1.0 - c["dc_loss1"]) * x[self.pdf] + x[self.pdt]
        L = self.lim
        out[self.r_thf] = x[self.pf[L]] ** 2 + x[self.qf[L]] ** 2
        out[self.r_tht] = x[self.pt[L]] ** 2 + x[self.qt[L]] ** 2
        nb = self.nb
        vm = x[self.vm]
        pb = np.bincount(c["gen_bus"], x[self.pg], nb) - np.bincount(f, x[self.pf], nb) - np.bincount(t, x[self.pt], nb) - c["gs"] * vm * vm
        qb = np.bincount(c["gen_bus"], x[self.qg], nb) - np.bincount(f, x[self.qf], nb) - np.bincount(t, x[self.qt], nb) + c["bs"] * vm * vm
        if nd:
            pb -= np.bincount(c["dc_f"], x[self.pdf], nb) + np.bincount(c["dc_t"], x[self.pdt], nb)
            qb -= np.bincount(c["dc_f"], x[self.qdf], nb) + np.bincount(c["dc_t"], x[self.qdt], nb)
        out[self.r_pb] = pb
        out[self.r_qb] = qb
        vf, vt, cs, sn, vv, pfr, qfr, pto, qto = self._flows(x)
        out[self.r_pfr] = x[self.pf] - pfr
        out[self.r_qfr] = x[self.qf] - qfr
        out[self.r_pto] = x[self.pt] - pto
        out[self.r_qto] = x[self.qt] - qto
        return out

    def eval_jac_g(self, x, v):
        c = self.c
        nl, nd = self.nl, self.nd
        one = np.ones(nl)
        parts = [one, -one, one, -one, np.ones(len(c["ref"]))]
        if nd:
            parts += [1.0 - c["dc_loss1"], np.ones(nd)]
        L = self.lim
        parts += [2 * x[self.pf[L]], 2 * x[self.qf[L]], 2 * x[self.pt[L]], 2 * x[self.qt[L]]]
        parts += [np.ones(self.ng), -one, -one, np.ones(self.ng), -one, -one]
        if nd:
            parts += [-np.ones(nd)] * 4
        vm_sh = x[self.vm[self.sh]]
        parts += [-2 * c["gs"][self.sh] * vm_sh, 2 * c["bs"][self.sh] * vm_sh]
        vf, vt, cs, sn, vv, pfr, qfr, pto, qto = self._flows(x)
        # d/d(vf), d/d(vt), d/d(va_f) (= -d/d(va_t)) of the four flows
        dp_f = (2 * self.k_ff_p * vf + self.a_f * vt * cs + self.b_f * vt * sn, self.a_f * vf * cs + self.b_f * vf * sn,
                -self.a_f * vv * sn + self.b_f * vv * cs)
        dq_f = (2 * self.k_ff_q * vf - self.b_f * vt * cs + self.a_f * vt * sn, -self.b_f * vf * cs + self.a_f * vf * sn,
                self.b_f * vv * sn + self.a_f * vv * cs)
        dp_t = (self.a_t * vt * cs - self.b_t * vt * sn, 2 * self.k_tt_p * vt + self.a_t * vf * cs - self.b_t * vf * sn,
                -self.a_t * vv * sn - self.b_t * vv * cs)
        dq_t = (-self.b_t * vt * cs - self.a_t * vt * sn, 2 * self.k_tt_q * vt - self.b_t * vf * cs - self.a_t * vf * sn,
                self.b_t * vv * sn - self.a_t * vv * cs)
        for dvf, dvt, dth in (dp_f, dq_f, dp_t, dq_t):
            parts += [one, -dvf, -dvt, -dth, dth]
        v[:] = np.concatenate([np.asarray(p, float).ravel() for p in parts])
        return v


def function_model(case):
    """The same polar ACOPF as a FunctionModel (moi_evaluator.py), i.e. the way the reference receives it: affine and quadratic
    scalar functions in the wrapper's six lists (angle differences linear <= / >=; reference angle, dc-line loss and the nodal
    balances of buses without shunt linear ==; thermal limits quadratic <=; balances of buses with a shunt quadratic ==) and
    Ohm's law as the NLP block (src/MOI_wrapper.jl:683-689).  Rows and pattern come out in the wrapper's order, which is a
    permutation of `AcopfModel`'s; the NLP block carries the parameters of its device kernel (csrc/asm_eval_kernels.hip.h)."""
    from .moi_evaluator import FunctionModel, ScalarFunction, NlpBlock
    mdl = AcopfModel(case)
    c = case
    V = lambda idx: int(idx) + 1                       # 1-based variable index
    fm = FunctionModel(mdl.n, mdl.x_L, mdl.x_U)
    fm.start = {j + 1: float(v) for j, v in enumerate(mdl.x0)}          # init_opf.jl midpoint start as VariablePrimalStart
    f, t = c["f_bus"], c["t_bus"]
    for l in range(mdl.nl):
        fm.add_constraint(ScalarFunction(0.0, [(1.0, V(mdl.va[f[l]])), (-1.0, V(mdl.va[t[l]]))]), "le", c["angmax"][l])
    for l in range(mdl.nl):
        fm.add_constraint(ScalarFunction(0.0, [(1.0, V(mdl.va[f[l]])), (-1.0, V(mdl.va[t[l]]))]), "ge", c["angmin"][l])
    for b in c["ref"]:
        fm.add_constraint(ScalarFunction(0.0, [(1.0, V(mdl.va[b]))]), "eq", 0.0)
    for k in range(mdl.nd):
        fm.add_constraint(ScalarFunction(0.0, [(1.0 - c["dc_loss1"][k], V(mdl.pdf[k])), (1.0, V(mdl.pdt[k]))]), "eq", c["dc_loss0"][k])
    for l in mdl.lim:
        fm.add_constraint(ScalarFunction(0.0, [], [(2.0, V(mdl.pf[l]), V(mdl.pf[l])), (2.0, V(mdl.qf[l]), V(mdl.qf[l]))]), "le", c["rate_a"][l] ** 2)
    for l in mdl.lim:
        fm.add_constraint(ScalarFunction(0.0, [], [(2.0, V(mdl.pt[l]), V(mdl.pt[l])), (2.0, V(mdl.qt[l]), V(mdl.qt[l]))]), "le", c["rate_a"][l] ** 2)
    gens_at = [[] for _ in range(mdl.nb)]
    for g, b in enumerate(c["gen_bus"]):
        gens_at[b].append(g)
    fr_at = [[] for _ in range(mdl.nb)]; to_at = [[] for _ in range(mdl.nb)]
    for l in range(mdl.nl):
        fr_at[f[l]].append(l); to_at[t[l]].append(l)
    dcf_at = [[] for _ in range(mdl.nb)]; dct_at = [[] for _ in range(mdl.nb)]
    for k in range(mdl.nd):
        dcf_at[c["dc_f"][k]].append(k); dct_at[c["dc_t"][k]].append(k)
    for (gv, fv, tv, dfv, dtv, shunt, rhs) in ((mdl.pg, mdl.pf, mdl.pt, mdl.pdf, mdl.pdt, -2.0 * c["gs"], c["pd"]),
                                               (mdl.qg, mdl.qf, mdl.qt, mdl.qdf, mdl.qdt, 2.0 * c["bs"], c["qd"])):
        for b in range(mdl.nb):
            aff = [(1.0, V(gv[g])) for g in gens_at[b]] + [(-1.0, V(fv[l])) for l in fr_at[b]] + [(-1.0, V(tv[l])) for l in to_at[b]]
            aff += [(-1.0, V(dfv[k])) for k in dcf_at[b]] + [(-1.0, V(dtv[k])) for k in dct_at[b]]
            quad = [(float(shunt[b]), V(mdl.vm[b]), V(mdl.vm[b]))] if (c["gs"][b] != 0 or c["bs"][b] != 0) else []
            fm.add_constraint(ScalarFunction(0.0, aff, quad), "eq", rhs[b])
    fm.objective = ScalarFunction(float(np.sum(c["cost0"])), [(float(c["cost1"][g]), V(mdl.pg[g])) for g in range(mdl.ng)],
                                  [(2.0 * float(c["cost2"][g]), V(mdl.pg[g]), V(mdl.pg[g])) for g in range(mdl.ng) if c["cost2"][g] != 0.0])
    # ---- NLP block: Ohm's law, rows pfr | qfr | pto | qto, pattern in 4 groups x 5 sub-blocks of n_branch
    nl = mdl.nl
    rows, cols = [], []
    for g, pv in enumerate((mdl.pf, mdl.qf, mdl.pt, mdl.qt)):
        r = np.arange(g * nl, (g + 1) * nl) + 1
        for cc in (pv, mdl.vm[f], mdl.vm[t], mdl.va[f], mdl.va[t]):
            rows.append(r); cols.append(np.asarray(cc) + 1)
    r0 = mdl.r_pfr[0]
    j0 = len(mdl.j_row) - 20 * nl

    def eval_g(x, out):
        full = mdl.eval_g(x, np.zeros(mdl.m))
        out[:] = full[r0:]
        return out

    def eval_jac_g(x, out):
        full = mdl.eval_jac_g(x, np.zeros(len(mdl.j_row)))
        out[:] = full[j0:]
        return out
    ipar = np.concatenate([[nl, mdl.va[0], mdl.vm[0], mdl.pf[0] if nl else 0, mdl.pt[0] if nl else 0, mdl.qf[0] if nl else 0, mdl.qt[0] if nl else 0],
                           f, t]).astype(np.int64)
    dpar = np.concatenate([mdl.k_ff_p, mdl.k_ff_q, mdl.k_tt_p, mdl.k_tt_q, mdl.a_f, mdl.b_f, mdl.a_t, mdl.b_t]).astype(np.float64)
    fm.nlp = NlpBlock(np.zeros(4 * nl), np.zeros(4 * nl), np.concatenate(rows), np.concatenate(cols), eval_g, eval_jac_g,
                      device=("acopf_ohm", ipar, dpar))
    fm.acopf_model = mdl
    return fm


def acopf_problem(case, name="acopf"):
    mdl = AcopfModel(case)
    pr = Problem(name, mdl.n, mdl.m, mdl.x_L, mdl.x_U, mdl.g_L, mdl.g_U, mdl.j_row, mdl.j_col, mdl.x0,
                 mdl.eval_f, mdl.eval_grad_f, mdl.eval_g, mdl.eval_jac_g)
    pr.model = mdl
    return pr


# element counts of the public MATPOWER cases the reference's configs name but does not ship
PUBLIC_CASE_SIZES = {"case118": (118, 54, 186), "case300": (300, 69, 411), "case1354pegase": (1354, 260, 1991)}


def synthetic_case(name, seed=1, load_scale=1.0):
    nb, ng, nl = PUBLIC_CASE_SIZES[name]
    return synthetic_grid(nb, ng, nl, seed, load_scale)


def scenario_case(base, scenario, lo=0.9, hi=1.1):
    """Scenario `scenario` of a batch (BASELINE.json configs[4]): every bus load of `base` scaled by an independent
    U(lo, hi) factor, seed = scenario id."""
    c = dict(base)
    f = lo + (hi - lo) * uniform01(1000 + int(scenario), base["n_bus"])
    c["pd"] = base["pd"] * f
    c["qd"] = base["qd"] * f
    return c
