"""`Parameters` with the reference's names and defaults (src/parameters.jl:1-29).  `external_optimizer`
keeps its role as the plugin slot (src/parameters.jl:7, consumed at src/algorithms/slp.jl:32): here it
is a factory `(data, j_row, j_col) -> AbstractSubOptimizer`; unset selects the HIP sub-optimizer."""
from dataclasses import dataclass
from typing import Any


@dataclass
class Parameters:
    mode: str = "Normal"
    method: str = "SLP"
    algorithm: str = "Line Search"
    external_optimizer: Any = None
    hessian_type: str = "none"
    OutputFlag: int = 0
    StatisticsFlag: int = 0
    tol_direction: float = 1.e-6
    tol_residual: float = 0.01
    tol_infeas: float = 0.01
    max_iter: int = 1000
    time_limit: float = float("inf")
    mu_merit: float = float("inf")
    max_mu: float = 1.e+6
    rho: float = 0.8
    eta: float = 0.4
    tau: float = 0.9
    min_alpha: float = 1.e-6
    tr_size: float = 0.4
    # not in the reference: evaluate f, grad f, g and the Jacobian values on the GPU (needs a Problem built from a FunctionModel,
    # activesetmethods_amd/moi_evaluator.py) and run the per-iteration norms / merit reductions there
    device_eval: bool = False


def get_parameter(params, pname):          # src/parameters.jl:31-33
    return getattr(params, pname)


def set_parameter(params, pname, val):     # src/parameters.jl:35-38
    if not hasattr(params, pname):
        raise AttributeError(pname)
    setattr(params, pname, val)
