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
