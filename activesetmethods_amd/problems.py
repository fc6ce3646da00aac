"""Problem definitions that feed the SLP hot path (inputs only - no solver logic lives here).

Each builder returns a `Problem` with the data the reference's `Model` (src/model.jl:1-61) receives
from the MOI wrapper (src/MOI_wrapper.jl:1014-1152): dimensions, bounds, the 1-based COO Jacobian
pattern `j_str` in the wrapper's row order (src/MOI_wrapper.jl:683-746), the start point
(src/MOI_wrapper.jl:1113-1130) and the four evaluation callbacks (src/MOI_wrapper.jl:1037-1069).

  toy_problem()            examples/toy_example.jl:12-18 == test/ext_solver.jl:12-18
  synthetic_dense_nlp()    BASELINE.json configs[1]  ("C2")
"""
import numpy as np

INF = np.inf
_MASK = (1 << 64) - 1


class Problem:
    def __init__(self, name, n, m, x_L, x_U, g_L, g_U, j_row, j_col, x0, eval_f, eval_grad_f, eval_g, eval_jac_g):
        self.name = name
        self.n, self.m = int(n), int(m)
        self.x_L, self.x_U = np.asarray(x_L, float), np.asarray(x_U, float)
        self.g_L, self.g_U = np.asarray(g_L, float), np.asarray(g_U, float)
        self.j_row = np.asarray(j_row, np.int64)      # 1-based, duplicates allowed
        self.j_col = np.asarray(j_col, np.int64)
        self.x0 = np.asarray(x0, float)
        self.eval_f, self.eval_grad_f, self.eval_g, self.eval_jac_g = eval_f, eval_grad_f, eval_g, eval_jac_g

    @property
    def nnz(self):
        return len(self.j_row)

    @property
    def j_str(self):
        return list(zip(self.j_row.tolist(), self.j_col.tolist()))


# --------------------------------------------------------------------------- toy (config C1)
def toy_problem():
    """min X^2 + X  s.t.  X >= -2 (affine row, first: src/MOI_wrapper.jl:683-689),
    X^2 - X == 2, X*Y == 1, X*Y >= 0 (NLP block).  Start (0,0).  Reference answer (-1,-1), LOCALLY_SOLVED
    (test/runtests.jl:11-13)."""
    def eval_f(x):
        return x[0] * x[0] + x[0]

    def eval_grad_f(x, g):
        g[0] = 2.0 * x[0] + 1.0
        g[1] = 0.0
        return g

    def eval_g(x, g):
        g[0] = x[0]
        g[1] = x[0] * x[0] - x[0]
        g[2] = x[0] * x[1]
        g[3] = x[0] * x[1]
        return g

    def eval_jac_g(x, v):
        v[0] = 1.0
        v[1] = 2.0 * x[0] - 1.0
        v[2] = x[1]
        v[3] = x[0]
        v[4] = x[1]
        v[5] = x[0]
        return v

    return Problem("toy", 2, 4, [-INF, -INF], [INF, INF], [-2.0, 2.0, 1.0, 0.0], [INF, 2.0, 1.0, INF],
                   [1, 2, 3, 3, 4, 4], [1, 1, 1, 2, 1, 2], [0.0, 0.0], eval_f, eval_grad_f, eval_g, eval_jac_g)


# --------------------------------------------------------------------------- synthetic dense NLP (config C2)
def splitmix64(seed, count):
    """`count` successive SplitMix64 outputs for `seed` (uint64 array)."""
    gamma = np.uint64(0x9E3779B97F4A7C15)
    with np.errstate(over='ignore'):
        z = np.uint64(seed & _MASK) + gamma * np.arange(1, count + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, count):
    """53-bit uniforms in (0,1): ((z >> 11) + 0.5) * 2^-53."""
    z = splitmix64(seed, count)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal01(seed, count):
    """Box-Muller on consecutive uniform pairs (cos branch then sin branch)."""
    k = (count + 1) // 2
    u = uniform01(seed, 2 * k)
    r = np.sqrt(-2.0 * np.log(u[0::2]))
    t = 2.0 * np.pi * u[1::2]
    out = np.empty(2 * k)
    out[0::2] = r * np.cos(t)
    out[1::2] = r * np.sin(t)
    return out[:count]


def synthetic_dense_nlp(n=1000, m=500):
    """g_i(x) = sum_j a_ij x_j + 1/2 q_ij x_j^2  (fully dense Jacobian J_ij = a_ij + q_ij x_j, row-major
    j_str); rows 0..m/2-1 equalities g_i = g_i(x*), the rest `<= g_i(x*) + 0.1`;
    f(x) = c'x + 1/2 sum d_j x_j^2; -1 <= x <= 1; x0 = 0.  Seeds: A 1, Q 2, x* 3, c 4, d 5."""
    A = normal01(1, m * n).reshape(m, n) / np.sqrt(n)
    Q = (0.1 * uniform01(2, m * n)).reshape(m, n) / n
    xs = uniform01(3, n) - 0.5
    c = normal01(4, n)
    d = 0.5 + uniform01(5, n)
    gs = A @ xs + 0.5 * (Q @ (xs * xs))
    h = m // 2
    g_L = np.full(m, -INF)
    g_U = np.full(m, INF)
    g_L[:h] = gs[:h]
    g_U[:h] = gs[:h]
    g_U[h:] = gs[h:] + 0.1
    j_row = np.repeat(np.arange(1, m + 1, dtype=np.int64), n)
    j_col = np.tile(np.arange(1, n + 1, dtype=np.int64), m)

    def eval_f(x):
        return float(c @ x + 0.5 * np.sum(d * x * x))

    def eval_grad_f(x, g):
        g[:] = c + d * x
        return g

    def eval_g(x, g):
        g[:] = A @ x + 0.5 * (Q @ (x * x))
        return g

    def eval_jac_g(x, v):
        v[:] = (A + Q * x).ravel()
        return v

    pr = Problem("synthetic_dense_n%d_m%d" % (n, m), n, m, -np.ones(n), np.ones(n), g_L, g_U, j_row, j_col,
                 np.zeros(n), eval_f, eval_grad_f, eval_g, eval_jac_g)
    pr.data = dict(A=A, Q=Q, c=c, d=d, xs=xs)
    return pr


def synthetic_dense_function_model(n=1000, m=500):
    """The synthetic dense NLP as a FunctionModel: quadratic objective in the affine / quadratic store, the dense rows as an NLP
    block with the `dense_quadratic` device kernel (csrc/asm_eval_kernels.hip.h).  Same problem as `synthetic_dense_nlp`."""
    from .moi_evaluator import FunctionModel, ScalarFunction, NlpBlock
    pr = synthetic_dense_nlp(n, m)
    d = pr.data
    fm = FunctionModel(n, pr.x_L, pr.x_U)
    fm.objective = ScalarFunction(0.0, [(float(d["c"][j]), j + 1) for j in range(n)], [(float(d["d"][j]), j + 1, j + 1) for j in range(n)])
    fm.nlp = NlpBlock(pr.g_L, pr.g_U, pr.j_row, pr.j_col, pr.eval_g, pr.eval_jac_g,
                      device=("dense_quadratic", np.zeros(1, np.int64), np.concatenate([d["A"].ravel(), d["Q"].ravel()])))
    return fm
