"""Python twin of `AsmHip.Optimizer <: MOI.AbstractOptimizer` (the Julia text) - the optimizer the
UNMODIFIED reference instantiates through `"external_optimizer" => AsmHip.Optimizer` (src/algorithms/slp.jl:32).

No Julia runtime exists in the build image, so the Julia module cannot be executed there.  This file restates it method by
method (same fields, same bookkeeping, same C calls) so that the part of the glue that is not the C ABI - recording the MOI
calls of `create_model!` / `sub_optimize!` (src/algorithms/subproblem.jl:51-215, 229-542) and answering the getters - is
exercised by tests: `tests/test_moi_replay_gpu.py` drives it with the reference's exact MOI call sequence (oracle/moi_replay.py)
and compares the result with `asm_sublp_solve` on the same sub-problem.

MOI types are represented by small records (only what the reference's calls carry); method names follow the MOI functions:
    MOI.empty!            -> empty()                 MOI.add_variables / add_variable -> add_variables(n) / add_variable()
    MOI.add_constraint    -> add_constraint(f, s)    MOI.set(ConstraintSet)  -> set_constraint_set(ci, s)
    MOI.transform         -> transform(ci, s)        MOI.modify(ci, change)  -> modify_constraint(ci, change)
    MOI.set(Objective..)  -> set_objective_function(f) / set_objective_sense(s) / modify_objective(change)
    MOI.optimize!         -> optimize()              MOI.get(...)            -> get_termination_status / get_variable_primal /
                                                                               get_constraint_dual
"""
from collections import namedtuple

import numpy as np

from .subproblem import HipSubOptimizer, QpData

# ---- the MOI records the reference passes (MOI 0.9 names)
VariableIndex = namedtuple("VariableIndex", "value")
SingleVariable = namedtuple("SingleVariable", "variable")
ScalarAffineTerm = namedtuple("ScalarAffineTerm", "coefficient variable_index")
ScalarAffineFunction = namedtuple("ScalarAffineFunction", "terms constant")
GreaterThan = namedtuple("GreaterThan", "lower")
LessThan = namedtuple("LessThan", "upper")
EqualTo = namedtuple("EqualTo", "value")
ScalarCoefficientChange = namedtuple("ScalarCoefficientChange", "variable new_coefficient")
ScalarConstantChange = namedtuple("ScalarConstantChange", "new_constant")
# ConstraintIndex{F,S}: kind = "SVG" / "SVL" / "SVE" (SingleVariable in GreaterThan / LessThan / EqualTo) or "AFF" (a row)
ConstraintIndex = namedtuple("ConstraintIndex", "kind value")

OPTIMIZE_NOT_CALLED, OPTIMAL, INFEASIBLE, DUAL_INFEASIBLE, OTHER_ERROR = "OPTIMIZE_NOT_CALLED", "OPTIMAL", "INFEASIBLE", "DUAL_INFEASIBLE", "OTHER_ERROR"
MIN_SENSE = "MIN_SENSE"


class UnsupportedError(Exception):
    """MOI.UnsupportedError: the optimizer is a sub-LP optimizer for this package, not a general LP solver."""


class Optimizer:
    """mutable struct Optimizer <: MOI.AbstractOptimizer (field for field)."""

    def __init__(self, device=0):
        self.device = device
        self.handle = None                      # HipSubOptimizer: asm_create + asm_sublp_setup + asm_lp_solve of one skeleton
        self.empty()

    # ---- model building: the calls of create_model! (subproblem.jl:51-215)
    def empty(self):                                                       # MOI.empty!, :54
        self.n = 0
        self.nslack = 0                          # slack columns in creation order: MOI variable index n + k, k = 1..nslack
        self.rtype, self.rhs, self.row_slacks = [], [], []                 # rows in creation order
        self.coef = {}                           # (row, column <= n) -> coefficient, from modify(ScalarCoefficientChange), :442-456
        self.q, self.w = np.zeros(0), []
        self.lb, self.ub = np.zeros(0), np.zeros(0)
        self.slo, self.sfixed = [], []
        self.skeleton_sent = False
        self.pattern = []
        self.status = OPTIMIZE_NOT_CALLED
        if self.handle is not None:
            self.handle.close()
            self.handle = None

    def is_empty(self):
        return self.n == 0

    def add_variables(self, n):                                           # :75
        if self.n != 0:
            raise UnsupportedError("AsmHip: one add_variables(n) per model")
        self.n = n
        self.q = np.zeros(n); self.lb = np.full(n, -np.inf); self.ub = np.full(n, np.inf)
        return [VariableIndex(i) for i in range(1, n + 1)]

    def add_variable(self):                                               # slack columns, :86-88
        self.nslack += 1
        self.w.append(0.0); self.slo.append(0.0); self.sfixed.append(False)
        return VariableIndex(self.n + self.nslack)

    def _isx(self, v):
        return v.value <= self.n

    def _sl(self, v):
        return v.value - self.n                  # 1-based position among the slack columns

    def add_constraint(self, f, s):
        if isinstance(f, SingleVariable):
            v = f.variable
            if isinstance(s, GreaterThan):                                # x: :133; slacks: :94-108
                if self._isx(v):
                    self.lb[v.value - 1] = s.lower
                else:
                    self.slo[self._sl(v) - 1] = s.lower; self.sfixed[self._sl(v) - 1] = False
                return ConstraintIndex("SVG", v.value)
            if isinstance(s, LessThan):                                   # :129
                if not self._isx(v):
                    raise UnsupportedError("AsmHip: slack columns have no upper bound")
                self.ub[v.value - 1] = s.upper
                return ConstraintIndex("SVL", v.value)
            raise UnsupportedError("AsmHip: SingleVariable in %r" % (s,))
        # rows: ScalarAffineFunction of slack terms only (:146-212)
        if any(self._isx(t.variable_index) for t in f.terms):
            raise UnsupportedError("AsmHip: rows are created with slack terms only")
        t, r = (0, s.value) if isinstance(s, EqualTo) else ((1, s.lower) if isinstance(s, GreaterThan) else (-1, s.upper))
        self.rtype.append(t); self.rhs.append(r - f.constant)
        self.row_slacks.append([(self._sl(term.variable_index), term.coefficient) for term in f.terms])
        self.skeleton_sent = False
        return ConstraintIndex("AFF", len(self.rtype))

    def set_constraint_set(self, c, s):                                   # MOI.set(ConstraintSet()), :307-374, 432-433, 468-483
        if c.kind == "SVG":
            if c.value <= self.n:
                self.lb[c.value - 1] = s.lower
            else:
                self.slo[c.value - self.n - 1] = s.lower; self.sfixed[c.value - self.n - 1] = False
        elif c.kind == "SVL":
            self.ub[c.value - 1] = s.upper
        elif c.kind == "SVE":
            self.slo[c.value - self.n - 1] = s.value; self.sfixed[c.value - self.n - 1] = True
        else:
            self.rhs[c.value - 1] = s.value if isinstance(s, EqualTo) else (s.lower if isinstance(s, GreaterThan) else s.upper)

    def transform(self, c, s):
        """slack s >= lo  <->  s == 0 (normal phase fixes every slack, :418-423; restoration frees them again, :301-367)"""
        k = c.value - self.n - 1
        if c.kind == "SVG" and isinstance(s, EqualTo):
            self.slo[k] = s.value; self.sfixed[k] = True
            return ConstraintIndex("SVE", c.value)
        if c.kind == "SVE" and isinstance(s, GreaterThan):
            self.slo[k] = s.lower; self.sfixed[k] = False
            return ConstraintIndex("SVG", c.value)
        if c.kind == "SVG" and isinstance(s, GreaterThan):
            self.slo[k] = s.lower
            return c
        if c.kind == "SVE" and isinstance(s, EqualTo):
            self.slo[k] = s.value
            return c
        raise UnsupportedError("AsmHip: transform %r -> %r" % (c, s))

    def modify_constraint(self, c, ch):                                   # :442-456
        if not self._isx(ch.variable):
            raise UnsupportedError("AsmHip: slack coefficients are fixed by the skeleton")
        key = (c.value, ch.variable.value)
        if key not in self.coef:
            self.skeleton_sent = False          # a new pattern entry: the skeleton is sent again
        self.coef[key] = ch.new_coefficient

    # objective (:115-120, 252-272, 385-405)
    def set_objective_function(self, f):
        self.q[:] = 0.0
        self.w = [0.0] * self.nslack
        for t in f.terms:
            if self._isx(t.variable_index):
                self.q[t.variable_index.value - 1] += t.coefficient
            else:
                self.w[self._sl(t.variable_index) - 1] += t.coefficient

    def set_objective_sense(self, s):
        if s != MIN_SENSE:
            raise UnsupportedError("AsmHip: MIN_SENSE (slp.jl:10)")

    def modify_objective(self, ch):
        if isinstance(ch, ScalarConstantChange):
            return                              # c0 does not move the optimum
        if self._isx(ch.variable):
            self.q[ch.variable.value - 1] = ch.new_coefficient
        else:
            self.w[self._sl(ch.variable) - 1] = ch.new_coefficient

    # ---- the skeleton in the vocabulary of asm_sublp_setup.  Every row the reference created is sent as a row of its own kind (EqualTo ->
    # equality with slacks +1 / -1, GreaterThan -> lower-only with slack +1, LessThan -> upper-only with slack -1): the extra `<=` row of a
    # range constraint (subproblem.jl:200-214) IS an upper-only row whose coefficients the reference sets itself (:448-457), so nothing has
    # to be recognised.  The library numbers slack columns row by row; `sperm[k]` = library column of the MOI slack k (creation order).
    def send_skeleton(self):
        R = len(self.rtype)
        lib_order = []
        for r in range(R):
            want = {0: [1.0, -1.0], 1: [1.0], -1: [-1.0]}[self.rtype[r]]
            if [c for _, c in self.row_slacks[r]] != want:
                raise UnsupportedError("AsmHip: row %d does not carry the slack terms create_model! gives its kind" % (r + 1))
            lib_order += [k for k, _ in self.row_slacks[r]]
        if sorted(lib_order) != list(range(1, self.nslack + 1)):
            raise UnsupportedError("AsmHip: every slack column must appear in exactly one row")
        self.sperm = np.empty(self.nslack, np.int64)
        self.sperm[np.array(lib_order, np.int64) - 1] = np.arange(self.nslack)
        c_lb = np.array([0.0 if t == 0 else (0.0 if t == 1 else -np.inf) for t in self.rtype])     # only the KIND of a row matters
        c_ub = np.array([0.0 if t == 0 else (np.inf if t == 1 else 0.0) for t in self.rtype])
        keys = sorted(self.coef)
        j_row = np.array([k[0] for k in keys], np.int64); j_col = np.array([k[1] for k in keys], np.int64)
        inf = np.full(self.n, np.inf)
        if self.handle is not None:
            self.handle.close()
        self.handle = HipSubOptimizer(QpData(np.zeros(self.n), 0.0, np.zeros(len(keys)), np.zeros(R), c_lb, c_ub, -inf, inf), j_row, j_col,
                                      device=self.device)
        self.skeleton_sent = True
        self.pattern = keys

    def optimize(self):                                                    # MOI.optimize!, :490
        if not self.skeleton_sent:
            self.send_skeleton()
        dE = np.array([self.coef[k] for k in self.pattern])
        use_slacks = not all(self.sfixed)
        w = np.zeros(self.nslack); slo = np.zeros(self.nslack)
        w[self.sperm] = self.w; slo[self.sperm] = self.slo                # MOI creation order -> library column order
        p, s, y, z, bstate, st = self.handle.lp_solve(dE, self.q, np.array(self.rhs), self.lb, self.ub, w if use_slacks else None,
                                                      slo if use_slacks else None)
        self.p, self.y, self.z, self.bstate = p, y, z, bstate
        self.s = s[self.sperm] if use_slacks else np.zeros(self.nslack)
        self.status = (OPTIMAL, INFEASIBLE, DUAL_INFEASIBLE, OTHER_ERROR)[st - 1]

    # ---- getters used at subproblem.jl:491-520
    def get_termination_status(self):
        return self.status

    def get_variable_primal(self, v):
        if not isinstance(v, VariableIndex):                  # vector form (subproblem.jl:502, 504)
            return [self.get_variable_primal(u) for u in v]
        return float(self.p[v.value - 1]) if self._isx(v) else float(self.s[self._sl(v) - 1])

    def get_constraint_dual(self, c):
        """MOI sign convention: >= rows / lower bounds >= 0, <= rows / upper bounds <= 0.  A fixed column (lb == ub) reports the
        two halves of its reduced cost like a simplex code does."""
        if not isinstance(c, ConstraintIndex):                # vector form (subproblem.jl:519-520)
            return [self.get_constraint_dual(u) for u in c]
        if c.kind == "AFF":
            return float(self.y[c.value - 1])
        j = c.value - 1
        if c.kind == "SVL":                                               # upper-bound dual (:519)
            return min(float(self.z[j]), 0.0) if (self.bstate[j] > 0 or self.lb[j] == self.ub[j]) else 0.0
        if c.kind == "SVG" and c.value <= self.n:                          # lower-bound dual (:520)
            return max(float(self.z[j]), 0.0) if (self.bstate[j] < 0 or self.lb[j] == self.ub[j]) else 0.0
        return 0.0

    def get_solver_name(self):
        return "asm-hip (MI355X)"

    def close(self):
        if self.handle is not None:
            self.handle.close()
            self.handle = None
