"""Scenario batches across GPUs (BASELINE.json configs[4]).

Independent NLP instances are the only data-parallel axis of the path: one SLP run is strictly
sequential.  Scenarios are block-partitioned over ranks (one process per GPU), every rank solves its
own block - `concurrency` scenarios at a time, each on its own handle / HIP stream (stream pool) - with no
data-path collective, and a single all-reduce of a
<= 8-double vector (RCCL over xGMI on the GPU box, gloo in the CPU tests) merges the convergence
statistics at the end: sum{#scenarios, #converged, SLP iterations, LP solves, restoration solves},
max{final inf_pr, final inf_du, wall seconds}."""
import time

import numpy as np

SUM_KEYS = ("scenarios", "converged", "iterations", "lp_solves", "restoration_solves")
MAX_KEYS = ("inf_pr", "inf_du", "wall_s")


def partition(n_items, world, rank):
    """Static block partition: items [lo, hi) of rank `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def local_stats(slps, wall_s):
    s = dict.fromkeys(SUM_KEYS + MAX_KEYS, 0.0)
    for slp in slps:
        s["scenarios"] += 1
        s["converged"] += 1 if slp.ret == 0 else 0
        s["iterations"] += slp.iter
        s["lp_solves"] += slp.lp_solves
        s["restoration_solves"] += sum(1 for r in slp.trace if r["fr"])
        s["inf_pr"] = max(s["inf_pr"], float(slp.prim_infeas) if np.isfinite(slp.prim_infeas) else 0.0)
        s["inf_du"] = max(s["inf_du"], float(slp.dual_infeas) if np.isfinite(slp.dual_infeas) else 0.0)
    s["wall_s"] = float(wall_s)
    return s


def reduce_stats(stats, device=None):
    """All-reduce the statistics over the default process group (no-op without one)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dict(stats)
    dev = device if device is not None else "cpu"
    sums = torch.tensor([stats[k] for k in SUM_KEYS], dtype=torch.float64, device=dev)
    maxs = torch.tensor([stats[k] for k in MAX_KEYS], dtype=torch.float64, device=dev)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(maxs, op=dist.ReduceOp.MAX)
    out = {k: float(v) for k, v in zip(SUM_KEYS, sums.tolist())}
    out.update({k: float(v) for k, v in zip(MAX_KEYS, maxs.tolist())})
    return out


def solve_batch(make_model, n_scenarios, rank=0, world=1, run=None, reduce_device=None, concurrency=1):
    """Solve scenarios [lo, hi) of this rank on this rank's GPU; `make_model(s)` returns the Model of scenario s,
    `run(model)` the finished SLP object (default: activesetmethods_amd.optimize).

    `concurrency` > 1 runs that many scenarios at a time, each on its own handle = its own HIP stream, from a pool of
    host threads (the C ABI is re-entrant per handle and releases the GIL): a single case300-sized SLP run is bound by
    the latency of its serial kernel chain and leaves most of the GPU idle, so independent runs overlap almost freely.
    Scenarios are handed to the next free worker in index order (dynamic assignment inside the rank); the results are
    returned in index order and do not depend on the concurrency."""
    if run is None:
        from .slp import optimize as run
    lo, hi = partition(n_scenarios, world, rank)
    t0 = time.perf_counter()
    if concurrency <= 1 or hi - lo <= 1:
        slps = [run(make_model(s)) for s in range(lo, hi)]
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(concurrency)) as pool:
            slps = list(pool.map(lambda s: run(make_model(s)), range(lo, hi)))
    st = local_stats(slps, time.perf_counter() - t0)
    return slps, reduce_stats(st, reduce_device)


# ------------------------------------------------------------------------------------------------------------------------------
# Lockstep scenario batch (include/asm_hip.h: asm_batch_*): B scenarios with the same pattern advance through ONE launch sequence
# on one stream, driven by one host thread inside the library (the native SLP driver asm_slp_run per scenario, as fibers).
# ------------------------------------------------------------------------------------------------------------------------------
def slp_params(par, max_lp_solves=0):
    """asm_slp_params from a `Parameters` object (src/parameters.jl:17-28)."""
    from . import _lib
    return _lib.SlpParams(int(par.max_iter), int(max_lp_solves or 0), float(par.tol_direction), float(par.tol_residual), float(par.tol_infeas),
                          float(par.eta), float(par.tau), float(par.min_alpha))


class NativeRun:
    """Outcome of one native SLP run (asm_slp_result + the final iterate): the attributes `local_stats` and the tests read from an SlpLS."""

    def __init__(self, res, x, lam, mult_x_U, mult_x_L, g):
        self.ret, self.iter, self.lp_solves = int(res.status), int(res.iter), int(res.lp_solves)
        self.restoration_solves, self.ls_trials, self.slot = int(res.restoration_solves), int(res.ls_trials), int(res.slot)
        self.paths = [int(v) for v in res.paths]
        self.ipm_iters, self.ns_cold = int(res.ipm_iters), int(res.ns_cold)
        self.obj_val, self.prim_infeas, self.dual_infeas, self.compl = float(res.obj_val), float(res.prim_infeas), float(res.dual_infeas), float(res.compl_)
        self.x, self.lam, self.mult_x_U, self.mult_x_L, self.E = x, lam, mult_x_U, mult_x_L, g
        self.trace = [dict(fr=True)] * self.restoration_solves       # (local_stats counts the restoration solves from the trace)


class HipBatch:
    """`n_slots` sub-problem handles with one LP skeleton and one evaluator on one device, advanced in lockstep (asm_batch_*).
    `problem`: a Problem built from a FunctionModel (its pattern, functions and NLP block are shared by every scenario; scenarios differ in
    bounds and start points)."""

    def __init__(self, problem, n_slots, device=0, groups=None):
        import ctypes as C
        from . import _lib
        from .subproblem import AsmHipError
        self._lib, self._C, self._err = _lib.load(), C, AsmHipError
        fm = getattr(problem, "function_model", None)
        if fm is None:
            raise ValueError("HipBatch needs a problem built from a FunctionModel (the batch evaluates on the device)")
        self.n, self.m, self.n_slots = int(problem.n), int(problem.m), int(n_slots)
        self._b = C.c_void_p()
        rc = self._lib.asm_batch_create(int(device), int(n_slots), C.byref(self._b))
        if rc != 0:
            raise AsmHipError("asm_batch_create(device=%d, n_slots=%d) failed with code %d" % (device, n_slots, rc))
        if groups is not None:
            self._check(self._lib.asm_batch_set_groups(self._b, int(groups)))
        self.groups = int(self._lib.asm_batch_groups(self._b))
        f64 = lambda a: np.ascontiguousarray(a, np.float64)
        jr, jc = np.ascontiguousarray(problem.j_row, np.int64), np.ascontiguousarray(problem.j_col, np.int64)
        gl, gu, xl, xu = map(f64, (problem.g_L, problem.g_U, problem.x_L, problem.x_U))
        self._check(self._lib.asm_batch_setup(self._b, self.n, self.m, len(jr), _lib.i64ptr(jr), _lib.i64ptr(jc), _lib.dptr(gl), _lib.dptr(gu),
                                              _lib.dptr(xl), _lib.dptr(xu)))
        fl = fm.flatten()
        kind, rows, nnz = 0, 0, 0
        ipar, dpar = np.zeros(1, np.int64), np.zeros(1)
        if fm.nlp is not None:
            name, ipar, dpar = fm.nlp.device
            kind = {"acopf_ohm": 1, "dense_quadratic": 2}[name]
            rows, nnz = fm.nlp.m, len(fm.nlp.rows)
            ipar, dpar = np.ascontiguousarray(ipar, np.int64), np.ascontiguousarray(dpar, np.float64)
        a = lambda k: fl[k]
        self._check(self._lib.asm_batch_eval_setup(self._b, fl["n_rows"], _lib.i64ptr(a("aff_ptr")), _lib.i64ptr(a("aff_var")), _lib.dptr(a("aff_coef")),
                                                   _lib.i64ptr(a("quad_ptr")), _lib.i64ptr(a("q_v1")), _lib.i64ptr(a("q_v2")), _lib.dptr(a("q_coef")),
                                                   _lib.dptr(a("constant")), _lib.i64ptr(a("jac_off")), _lib.i64ptr(a("g_ptr")), _lib.i64ptr(a("g_kind")),
                                                   _lib.dptr(a("g_coef")), _lib.i64ptr(a("g_other")), float(fl["objective_scale"]), kind, rows, nnz,
                                                   _lib.i64ptr(ipar), len(ipar) if kind else 0, _lib.dptr(dpar), len(dpar) if kind else 0))

    def _check(self, rc):
        if rc != 0:
            raise self._err("libasmhip batch error %d: %s" % (rc, self._lib.asm_batch_last_error(self._b).decode()))

    def close(self):
        if getattr(self, "_b", None):
            self._lib.asm_batch_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ns_basis(self):
        from . import _lib
        k = self._C.c_int64(0)
        self._check(self._lib.asm_batch_ns_basis(self._b, None, self._C.byref(k)))
        J = np.zeros(max(int(k.value), 1), np.int32)
        self._check(self._lib.asm_batch_ns_basis(self._b, _lib.i32ptr(J), self._C.byref(k)))
        return J[:int(k.value)]

    def set_ns_basis(self, J):
        from . import _lib
        J = np.ascontiguousarray(J, np.int32)
        self._check(self._lib.asm_batch_set_ns_basis(self._b, _lib.i32ptr(J), len(J)))

    def stats(self):
        from . import _lib
        s = _lib.BatchStats()
        self._check(self._lib.asm_batch_get_stats(self._b, self._C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def slp_run(self, g_L, g_U, x_L, x_U, x0, parameters, max_lp_solves=0):
        """Complete Line-Search SLP runs of `len(g_L)` scenarios (rows of the 2-D arrays): list of NativeRun in scenario order."""
        from . import _lib
        f64 = lambda a_: np.ascontiguousarray(a_, np.float64)
        g_L, g_U, x_L, x_U, x0 = map(f64, (g_L, g_U, x_L, x_U, x0))
        S = x0.shape[0]
        assert g_L.shape == (S, self.m) and g_U.shape == (S, self.m) and x_L.shape == (S, self.n) and x_U.shape == (S, self.n) and x0.shape == (S, self.n)
        if parameters.algorithm != "Line Search":
            raise ValueError("the native driver restates run!(::SlpLS) only")
        par = slp_params(parameters, max_lp_solves)
        x = np.empty((S, self.n)); lam = np.empty((S, max(self.m, 1))); mU = np.empty((S, self.n)); mL = np.empty((S, self.n)); g = np.empty((S, max(self.m, 1)))
        res = (_lib.SlpResult * S)()
        self._check(self._lib.asm_batch_slp_run(self._b, S, _lib.dptr(g_L), _lib.dptr(g_U), _lib.dptr(x_L), _lib.dptr(x_U), _lib.dptr(x0), self._C.byref(par),
                                                _lib.dptr(x), _lib.dptr(lam), _lib.dptr(mU), _lib.dptr(mL), _lib.dptr(g), res))
        return [NativeRun(res[s], x[s], lam[s, :self.m], mU[s], mL[s], g[s, :self.m]) for s in range(S)]

    def sublp_solve(self, dE, df, f, E, x_k, delta, feasibility, bounds=None):
        """asm_sublp_solve for `count` = len(f) <= n_slots scenarios in lockstep; `bounds` = (g_L, g_U, x_L, x_U) per scenario or None.
        Returns (p, lambda, mult_x_U, mult_x_L, p_slack, status) with a leading scenario dimension."""
        from . import _lib
        f64 = lambda a_: np.ascontiguousarray(a_, np.float64)
        dE, df, f, E, x_k, delta = map(f64, (dE, df, f, E, x_k, delta))
        cnt = len(f)
        fe = np.ascontiguousarray(feasibility, np.int32)
        p = np.empty((cnt, self.n)); lam = np.empty((cnt, max(self.m, 1))); mU = np.empty((cnt, self.n)); mL = np.empty((cnt, self.n))
        ps = np.empty((cnt, 2 * max(self.m, 1))); st = np.zeros(cnt, np.int32)
        if bounds is not None:
            bl = [f64(b_) for b_ in bounds]
            bp = [_lib.dptr(b_) for b_ in bl]
        else:
            bp = [None] * 4
        self._check(self._lib.asm_batch_sublp_solve(self._b, cnt, bp[0], bp[1], bp[2], bp[3], _lib.dptr(dE), _lib.dptr(df), _lib.dptr(f), _lib.dptr(E),
                                                    _lib.dptr(x_k), _lib.dptr(delta), _lib.i32ptr(fe), _lib.dptr(p), _lib.dptr(lam), _lib.dptr(mU), _lib.dptr(mL),
                                                    _lib.dptr(ps), _lib.i32ptr(st)))
        return p, lam[:, :self.m], mU, mL, ps[:, :2 * self.m], st

    def slot_stats(self, slot):
        from . import _lib
        s = _lib.SolveStats()
        h = self._lib.asm_batch_handle(self._b, int(slot))
        rc = self._lib.asm_sublp_last_stats(h, self._C.byref(s))
        if rc != 0:
            raise self._err("asm_sublp_last_stats(slot %d): %d" % (slot, rc))
        return {k: getattr(s, k) for k, _ in s._fields_}


def solve_batch_lockstep(problems, parameters, n_slots, device=0, rank=0, world=1, reduce_device=None, batch=None):
    """The scenarios of this rank (`problems`: list of Problems with one pattern, built from FunctionModels) through a HipBatch of `n_slots`
    slots; returns (runs, reduced statistics, batch statistics) like `solve_batch`.  `batch`: an existing HipBatch to re-use."""
    own = batch is None
    if own:
        batch = HipBatch(problems[0], min(int(n_slots), len(problems)), device)
    t0 = time.perf_counter()
    runs = batch.slp_run(np.stack([p.g_L for p in problems]), np.stack([p.g_U for p in problems]), np.stack([p.x_L for p in problems]),
                         np.stack([p.x_U for p in problems]), np.stack([p.x0 for p in problems]), parameters)
    st = local_stats(runs, time.perf_counter() - t0)
    bst = batch.stats()
    if own:
        batch.close()
    return runs, reduce_stats(st, reduce_device), bst


# ------------------------------------------------------------------------------------------------------------------------------
# Dynamic assignment across ranks ("consider dynamic work assignment via a host-side counter"): instead of the
# static block partition every rank claims chunks of scenario indices from one counter in the process group's key-value store.  It pays
# when a rank has more scenarios than batch slots AND the iteration counts differ between ranks; with as many slots as scenarios per
# GPU (the default shape: 64 / 64) every scenario of a rank runs concurrently and a rank's time is its slowest scenario - nothing to steal.
# ------------------------------------------------------------------------------------------------------------------------------
def claim_chunks(total, chunk, store=None, key="asm_batch_next"):
    """Yield (lo, hi) ranges of [0, total) claimed from a counter shared by the ranks: `store.add(key, chunk)` is atomic across processes
    (torch.distributed TCPStore / the default group's store).  Without a store: one local counter (single process)."""
    total, chunk = int(total), max(1, int(chunk))
    if store is None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            from torch.distributed import distributed_c10d
            store = distributed_c10d._get_default_store()
    local = [0]
    while True:
        if store is not None:
            hi = int(store.add(key, chunk))
        else:
            local[0] += chunk
            hi = local[0]
        lo = hi - chunk
        if lo >= total:
            return
        yield lo, min(hi, total)


def solve_batch_dynamic(problem_of, total, parameters, n_slots, batch, chunk=None, store=None, reduce_device=None):
    """Lockstep batch with dynamic assignment: this rank claims chunks of `chunk` scenarios (default: n_slots) until the counter runs out;
    `problem_of(s)` returns the Problem of scenario s.  Returns ({scenario: NativeRun}, reduced statistics)."""
    chunk = int(chunk or n_slots)
    t0 = time.perf_counter()
    mine = {}
    for lo, hi in claim_chunks(total, chunk, store):
        prs = [problem_of(s) for s in range(lo, hi)]
        runs = batch.slp_run(np.stack([p.g_L for p in prs]), np.stack([p.g_U for p in prs]), np.stack([p.x_L for p in prs]),
                             np.stack([p.x_U for p in prs]), np.stack([p.x0 for p in prs]), parameters)
        mine.update({lo + k: r for k, r in enumerate(runs)})
    st = local_stats(list(mine.values()), time.perf_counter() - t0)
    return mine, reduce_stats(st, reduce_device)
