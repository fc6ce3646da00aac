"""The evaluator block of the reference's MOI wrapper (src/MOI_wrapper.jl:683-1012), restated: the piece that turns a model
made of affine / quadratic scalar functions plus an optional NLP block into the four callbacks and the COO pattern `j_str`
the SLP hot path consumes.

    ScalarFunction                MOI.ScalarAffineFunction / MOI.ScalarQuadraticFunction (constant, affine terms, quadratic terms)
    FunctionModel                 the six constraint lists, objective, sense and NLP block of `Optimizer` (MOI_wrapper.jl:22-54)
      row offsets                 linear <=, >=, ==, quadratic <=, >=, ==, then the NLP block        (:683-689)
      jacobian_structure()        :693-746   quadratic terms emit (row, v1) and, if v1 != v2, (row, v2): duplicates possible
      eval_function / fill_gradient / eval_constraint / fill_constraint_jacobian / eval_constraint_jacobian   :776-944
      constraint_bounds()         :980-1012
      objective scale             MIN +1, MAX -1, FEASIBILITY 0 on f and its gradient                 (:1037-1054)
      start point                 user start, else 0 projected onto the bounds                        (:1113-1130)

Variables are 1-based in the public interface exactly as `VariableIndex.value`; the arithmetic follows the reference term
by term and in its order (`function_value += coefficient * x[...]`), which is what makes the device kernels of
csrc/asm_eval_kernels.hip.h (same order, no fused multiply-add) bit-identical to this host evaluator.
"""
import numpy as np

from .problems import Problem

INF = np.inf


class ScalarFunction:
    """constant + sum coef * x[var] + sum coef * x[v1] * x[v2]   (diagonal quadratic terms carry 1/2: MOI convention,
    MOI_wrapper.jl:791-797)."""

    def __init__(self, constant=0.0, affine=(), quadratic=()):
        self.constant = float(constant)
        self.affine = [(float(c), int(v)) for c, v in affine]                 # (coefficient, variable), variable 1-based
        self.quadratic = [(float(c), int(a), int(b)) for c, a, b in quadratic]  # (coefficient, variable_index_1, variable_index_2)

    # eval_function, MOI_wrapper.jl:780-807
    def value(self, x):
        v = self.constant
        for c, j in self.affine:
            v = v + c * x[j - 1]
        for c, a, b in self.quadratic:
            if a == b:
                v = v + 0.5 * c * x[a - 1] * x[b - 1]
            else:
                v = v + c * x[a - 1] * x[b - 1]
        return v

    # fill_gradient!, MOI_wrapper.jl:827-850 (grad is zeroed by the caller)
    def add_gradient(self, grad, x):
        for c, j in self.affine:
            grad[j - 1] += c
        for c, a, b in self.quadratic:
            if a == b:
                grad[a - 1] += c * x[a - 1]
            else:
                grad[a - 1] += c * x[b - 1]
                grad[b - 1] += c * x[a - 1]

    # append_to_jacobian_sparsity!, MOI_wrapper.jl:693-712
    def jacobian_columns(self):
        cols = [j for _, j in self.affine]
        for _, a, b in self.quadratic:
            cols.append(a)
            if a != b:
                cols.append(b)
        return cols

    # fill_constraint_jacobian!, MOI_wrapper.jl:889-918
    def jacobian_values(self, x):
        vals = [c for c, _ in self.affine]
        for c, a, b in self.quadratic:
            if a == b:
                vals.append(c * x[b - 1])
            else:
                vals.append(c * x[b - 1])
                vals.append(c * x[a - 1])
        return vals


class NlpBlock:
    """The `nlp_data` of the wrapper: bounds, pattern (1-based rows inside the block, 1-based columns) and callbacks
    eval_g(x, out) / eval_jac_g(x, out) in the pattern's order; `device` names a device kernel of libasmhip
    (None: host only) with its parameter arrays."""

    def __init__(self, g_L, g_U, rows, cols, eval_g, eval_jac_g, device=None):
        self.g_L, self.g_U = np.asarray(g_L, float), np.asarray(g_U, float)
        self.rows, self.cols = np.asarray(rows, np.int64), np.asarray(cols, np.int64)
        self.eval_g, self.eval_jac_g = eval_g, eval_jac_g
        self.device = device

    @property
    def m(self):
        return len(self.g_L)


class FunctionModel:
    def __init__(self, n, x_L=None, x_U=None):
        self.n = int(n)
        self.x_L = np.full(n, -INF) if x_L is None else np.asarray(x_L, float)
        self.x_U = np.full(n, INF) if x_U is None else np.asarray(x_U, float)
        self.start = {}                                   # VariablePrimalStart (1-based variable -> value)
        self.linear_le, self.linear_ge, self.linear_eq = [], [], []
        self.quadratic_le, self.quadratic_ge, self.quadratic_eq = [], [], []
        self.objective = None                             # ScalarFunction
        self.sense = "MIN_SENSE"
        self.nlp = None

    # ---- model building (the wrapper stores (func, set) pairs per list, MOI_wrapper.jl:14-20, 417-681)
    def add_constraint(self, func, kind, bound):
        quad = len(func.quadratic) > 0
        lst = {"le": self.quadratic_le if quad else self.linear_le, "ge": self.quadratic_ge if quad else self.linear_ge,
               "eq": self.quadratic_eq if quad else self.linear_eq}[kind]
        lst.append((func, float(bound)))
        return len(lst)

    def _lists(self):
        return (("le", self.linear_le), ("ge", self.linear_ge), ("eq", self.linear_eq),
                ("le", self.quadratic_le), ("ge", self.quadratic_ge), ("eq", self.quadratic_eq))

    def functions(self):
        return [f for _, lst in self._lists() for f, _ in lst]

    @property
    def nlp_constraint_offset(self):                      # MOI_wrapper.jl:689
        return sum(len(lst) for _, lst in self._lists())

    @property
    def m(self):
        return self.nlp_constraint_offset + (self.nlp.m if self.nlp is not None else 0)

    @property
    def objective_scale(self):                            # MOI_wrapper.jl:1037-1045
        return {"MIN_SENSE": 1.0, "MAX_SENSE": -1.0, "FEASIBILITY_SENSE": 0.0}[self.sense]

    # ---- MOI_wrapper.jl:726-746
    def jacobian_structure(self):
        j_str = []
        row = 1
        for f in self.functions():
            j_str += [(row, c) for c in f.jacobian_columns()]
            row += 1
        if self.nlp is not None:
            j_str += [(int(r) + row - 1, int(c)) for r, c in zip(self.nlp.rows, self.nlp.cols)]
        return j_str

    # ---- MOI_wrapper.jl:980-1012
    def constraint_bounds(self):
        lb, ub = [], []
        for kind, lst in self._lists():
            for _, b in lst:
                lb.append(-INF if kind == "le" else b)
                ub.append(INF if kind == "ge" else b)
        if self.nlp is not None:
            lb += list(self.nlp.g_L)
            ub += list(self.nlp.g_U)
        return np.array(lb, float), np.array(ub, float)

    # ---- callbacks (MOI_wrapper.jl:1046-1069 with eval_objective :809-820, eval_objective_gradient :852-861,
    #      eval_constraint :875-887, eval_constraint_jacobian :932-944)
    def eval_f(self, x):
        return self.objective_scale * (self.objective.value(x) if self.objective is not None else 0.0)

    def eval_grad_f(self, x, grad):
        grad[:] = 0.0
        if self.objective is not None:
            self.objective.add_gradient(grad, x)
        grad *= self.objective_scale
        return grad

    def eval_g(self, x, g):
        row = 0
        for f in self.functions():
            g[row] = f.value(x)
            row += 1
        if self.nlp is not None:
            self.nlp.eval_g(x, g[row:])
        return g

    def eval_jac_g(self, x, values):
        off = 0
        for f in self.functions():
            v = f.jacobian_values(x)
            values[off:off + len(v)] = v
            off += len(v)
        if self.nlp is not None:
            self.nlp.eval_jac_g(x, values[off:])
        return values

    # ---- start point (MOI_wrapper.jl:1113-1130): user start, else 0 projected onto the bounds
    def start_point(self):
        x0 = np.minimum(np.maximum(np.zeros(self.n), self.x_L), self.x_U)
        for j, v in self.start.items():
            x0[j - 1] = v
        return x0

    def to_problem(self, name="function_model"):
        j_str = self.jacobian_structure()
        g_L, g_U = self.constraint_bounds()
        pr = Problem(name, self.n, self.m, self.x_L, self.x_U, g_L, g_U, [r for r, _ in j_str], [c for _, c in j_str], self.start_point(),
                     self.eval_f, self.eval_grad_f, self.eval_g, self.eval_jac_g)
        pr.function_model = self
        return pr

    # ---- flattened store for the device evaluator (include/asm_hip.h: asm_eval_setup_functions)
    def flatten(self):
        """CSR-like arrays over the rows [constraint functions..., objective]: affine terms, quadratic terms, constants, the
        offset of each row's Jacobian values in `dE`, and per-variable contribution lists of the objective gradient in term
        order (kind 0: += coef ; 1: += coef * x[other])."""
        fs = self.functions()
        rows = fs + [self.objective if self.objective is not None else ScalarFunction()]
        aff_ptr, quad_ptr, jac_off = [0], [0], [0]
        aff_var, aff_coef, q1, q2, qc, const = [], [], [], [], [], []
        for f in rows:
            aff_var += [j - 1 for _, j in f.affine]; aff_coef += [c for c, _ in f.affine]
            q1 += [a - 1 for _, a, _ in f.quadratic]; q2 += [b - 1 for _, _, b in f.quadratic]; qc += [c for c, _, _ in f.quadratic]
            aff_ptr.append(len(aff_var)); quad_ptr.append(len(qc)); const.append(f.constant)
            jac_off.append(jac_off[-1] + len(f.jacobian_columns()))
        contrib = [[] for _ in range(self.n)]
        obj = rows[-1]
        for c, j in obj.affine:
            contrib[j - 1].append((0, c, 0))
        for c, a, b in obj.quadratic:
            if a == b:
                contrib[a - 1].append((1, c, a - 1))
            else:
                contrib[a - 1].append((1, c, b - 1))
                contrib[b - 1].append((1, c, a - 1))
        g_ptr = [0]
        g_kind, g_coef, g_other = [], [], []
        for lst in contrib:
            for k, c, o in lst:
                g_kind.append(k); g_coef.append(c); g_other.append(o)
            g_ptr.append(len(g_kind))
        i64 = lambda a: np.asarray(a, np.int64)
        f64 = lambda a: np.asarray(a, np.float64)
        return dict(n_rows=len(fs), aff_ptr=i64(aff_ptr), aff_var=i64(aff_var), aff_coef=f64(aff_coef), quad_ptr=i64(quad_ptr),
                    q_v1=i64(q1), q_v2=i64(q2), q_coef=f64(qc), constant=f64(const), jac_off=i64(jac_off[:len(fs) + 1]),
                    g_ptr=i64(g_ptr), g_kind=i64(g_kind), g_coef=f64(g_coef), g_other=i64(g_other),
                    objective_scale=self.objective_scale, nnz_functions=jac_off[len(fs)])
