"""asm-hip: MI355X-native sub-problem solver for sequential linear programming - a drop-in for the
per-iteration sub-LP path of exanauts/ActiveSetMethods.

The compute path is libasmhip.so (hand-written HIP for gfx950 behind the C ABI of include/asm_hip.h);
this package is the host-side mirror of the reference's interface for that path."""
from .parameters import Parameters, get_parameter, set_parameter
from .status import ApplicationReturnStatus
from .subproblem import QpData, HipSubOptimizer, AsmHipError
from .slp import Model, SlpLS, SlpTR, optimize
from . import problems

__all__ = ["Parameters", "get_parameter", "set_parameter", "ApplicationReturnStatus", "QpData", "HipSubOptimizer",
           "AsmHipError", "Model", "SlpLS", "SlpTR", "optimize", "problems"]
