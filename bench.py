#!/usr/bin/env python
"""Benchmark of the SLP sub-problem hot path (BASELINE.json metric: SLP iterations/sec).

  python bench.py --gpus N --steps K --warmup W [--workload c2|c2small] [--algorithm "Trust Region"]

N = 1 (default workload c4): a "step" is one SLP outer iteration that reaches the LP solve:
evaluation, Jacobian assembly, LP formulation, the HIP LP solve and the merit/step logic.
N > 1 (default workload c5, BASELINE.json configs[4] / the "batch-NLP solves/sec @8 GPU" half of the metric): scenario
ACOPF instances are block-partitioned over the ranks, a "step" is one complete scenario solve per GPU (default 64 per GPU =
512 over 8), no data-path collective, one all-reduce of the convergence statistics.  An explicit --workload c2|c3|c4 at
N > 1 runs independent replicas (the path does not shard inside one NLP - "replicas only").
value = work of all ranks / max-over-ranks time.  One JSON line is printed by rank 0.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one child process per
GPU, spawned before anything touches the GPU); under `python -m torch.distributed.run` the ranks already exist and are used.

The dominant kernel's roofline numbers come from HIP events recorded on the solver's own stream
around each launch (include/asm_hip.h: asm_kernel_stats_get); the CPU baseline is the oracle's NumPy
restatement of the same path (oracle/, "port") on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time


def _host_cpu_quota():
    """CPUs this process may actually use: the cgroup quota (cpu.max) if there is one, else the CPU count.  The GPU box
    exposes 256 logical CPUs but grants 16 per 100 ms period; BLAS/OpenMP pools sized by the CPU count spin
    through the quota and the kernel then freezes the whole process (incl. the HIP launch thread) for the rest
    of every period - measured as 60 ms stalls every 100 ms in the rocprof timeline."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except Exception:
        pass
    return os.cpu_count() or 1


def _cpu_throttled_s():
    """Seconds this cgroup has been frozen for exceeding its CPU quota so far (0 when unknown)."""
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            if line.startswith("throttled_usec"):
                return int(line.split()[1]) * 1e-6
    except Exception:
        pass
    return 0.0


# half of this rank's share of the quota: BLAS workers spin-wait after every call (ranks of one node share the cgroup)
HOST_THREADS = max(1, min(8, _host_cpu_quota() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))) // 2))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(HOST_THREADS))

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# dense FP64 matrix-core peak of MI355X (AMD public specification; the HIP guides in this image give no
# FP64 figure) and HBM3E peak (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
FP64_MFMA_PEAK_TFLOPS = 78.6
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (description, default algorithm, default steps, default warmup)
    "c4": dict(desc="ACOPF case1354pegase-sized synthetic grid (1354 bus / 260 gen / 1991 branch; BASELINE.json configs[3], "
                    "the case the metric is quoted on; real case file not shipped with the reference), load scale 0.5 - the instance whose "
                    "Line-Search SLP run terminates (status 0 after 76 iterations, all LPs in the normal phase), n=11192 m=18637",
               algorithm="Line Search", steps=20, warmup=2, case="case1354pegase", load_scale=0.5),
    "c4fr": dict(desc="the same grid at the nominal synthetic load (scale 1.0): the SLP run never leaves feasibility restoration - every "
                      "timed step is a restoration LP (min sum of slacks, non-unique optimum); round-1/2 headline workload, n=11192 m=18637",
                 algorithm="Line Search", steps=20, warmup=2, case="case1354pegase", load_scale=1.0),
    "c5": dict(desc="batch of scenario ACOPF, case300-sized synthetic grid (300 bus / 69 gen / 411 branch), load_scale 0.5, loads x U(0.9,1.1) per "
                    "scenario (BASELINE.json configs[4] has 512 scenarios over 8 GPUs = 64 per GPU; --scenarios-per-gpu sets the share), "
                    "n=2382 m=3889; one step = one complete scenario solve",
               algorithm="Line Search", steps=64, warmup=1),
    "c3": dict(desc="ACOPF case118-sized synthetic grid (118 bus / 54 gen / 186 branch; BASELINE.json configs[2]), n=1088 m=1725",
               algorithm="Line Search", steps=10, warmup=2, case="case118", load_scale=1.0),
    "c2": dict(desc="synthetic dense NLP n=1000 m=500 (BASELINE.json configs[1])", algorithm="Trust Region", steps=20, warmup=3),
    "c2small": dict(desc="synthetic dense NLP n=200 m=100 (reduced; parity-test size)", algorithm="Trust Region", steps=20, warmup=3),
}


def run_batch(args, rank, world, local_rank, dist, torch):
    """Workload c5: scenarios are block-partitioned over ranks (activesetmethods_amd.batch), every rank solves its share on its own GPU
    as ONE lockstep batch (include/asm_hip.h: asm_batch_*): `--slots` scenarios advance through one launch sequence on one stream, driven
    by one host thread inside the library (native SLP driver per scenario, launches of different scenarios merged into one launch with the
    scenario index in the grid); a slot takes the next scenario when it finishes one.  No data-path collective; one all-reduce merges
    the statistics.  `--batch-mode pool` keeps the round-3 path (host threads, one handle / HIP stream each, Python SLP driver)."""
    import numpy as np
    import activesetmethods_amd as A
    from activesetmethods_amd import acopf, batch
    base = acopf.synthetic_case("case300", 1, 0.5)     # half the nominal synthetic load: Line-Search SLP converges in ~60 iterations
    per_gpu = args.steps
    total = per_gpu * world
    lo, hi = batch.partition(total, world, rank)
    par = A.Parameters(algorithm=args.algorithm, max_iter=args.max_iter, device_eval=True)
    # the scenario NLPs of this rank as host objects (function lists, bounds, start points) before the timed region: inputs are in place
    # when the clock starts, as for the other workloads; bounds uploads (per scenario) and every solve are inside it
    own = range(0, total) if args.dynamic else range(lo, hi)      # dynamic assignment: any rank may claim any scenario
    problems = {s: acopf.function_model(acopf.scenario_case(base, s)).to_problem("case300-sized scenario %d" % s) for s in own}
    if args.batch_mode == "pool":
        return run_batch_pool(args, rank, world, local_rank, dist, torch, base, problems, total, per_gpu)
    n_slots = max(1, min(args.slots, hi - lo))
    groups = args.groups
    if groups is None:
        # one host thread per group: never more groups than this rank's share of the CPU quota (a spinning thread that is descheduled
        # stalls its whole group)
        share = max(1, _host_cpu_quota() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
        groups = max(1, min(3 if n_slots >= 48 else (2 if n_slots >= 16 else 1), share - 1 if share > 1 else 1))
    hb = batch.HipBatch(problems[lo], n_slots, device=local_rank, groups=groups)
    if args.warmup:
        # every slot allocates its null-space buffers in its first LP and the batch selects its reference basis columns: outside the clock
        wpr = [acopf.function_model(acopf.scenario_case(base, 10 ** 6 + rank * 4096 + k)).to_problem("warm-up") for k in range(n_slots)]
        hb.slp_run(np.stack([p.g_L for p in wpr]), np.stack([p.g_U for p in wpr]), np.stack([p.x_L for p in wpr]), np.stack([p.x_U for p in wpr]),
                   np.stack([p.x0 for p in wpr]), par, max_lp_solves=2 * args.warmup)
    st0 = hb.stats()
    plist = [problems[s] for s in range(lo, hi)]
    _freeze_heap()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    thr0 = _cpu_throttled_s()
    t0 = time.perf_counter()
    if args.dynamic:
        # chunks of n_slots scenarios claimed from a counter in the process group's store (activesetmethods_amd.batch.claim_chunks)
        mine, stats = batch.solve_batch_dynamic(problems.__getitem__, total, par, n_slots, hb, reduce_device=args.reduce_device)
        runs, st1 = list(mine.values()), hb.stats()
    else:
        runs, stats, st1 = batch.solve_batch_lockstep(plist, par, n_slots, rank=rank, world=world, reduce_device=args.reduce_device, batch=hb)
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    thr = _cpu_throttled_s() - thr0
    rank_s, rank_thr = [mine], [thr]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=args.reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        g = [torch.zeros(2, dtype=torch.float64, device=args.reduce_device) for _ in range(world)]
        dist.all_gather(g, torch.tensor([mine, thr], dtype=torch.float64, device=args.reduce_device))
        rank_s, rank_thr = [float(v[0]) for v in g], [float(v[1]) for v in g]
    if rank == 0:
        names = {0: "warm", 1: "ipm0+ln", 2: "ipm1+ln", 3: "ipm2+ln", 4: "ipm+face", 5: "ipm-unpolished", 6: "ipm-infeasible",
                 7: "phase1-infeasible", 8: "ipm~+ln", 9: "ipm+ref", 10: "ipm-conv"}
        hist = {}
        for r in runs:
            for k, c in enumerate(r.paths):
                if c:
                    hist[names.get(k, str(k))] = hist.get(names.get(k, str(k)), 0) + c
        d = {k: st1[k] - st0[k] for k in st1}
        out = {"metric": "batch-NLP solves/sec", "value": total / elapsed, "unit": "solves/s", "n_gpus": world, "steps": per_gpu,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / per_gpu, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": WORKLOADS["c5"]["desc"], "algorithm": args.algorithm, "max_iter": args.max_iter,
                          "scenarios_total": total, "assignment": "dynamic (chunks claimed from a shared counter)" if args.dynamic else "static block partition",
                          "parallelism": "scenarios block-partitioned, %d per GPU; per GPU ONE lockstep batch of %d slots in %d groups (one stream + one host thread each; "
                                         "asm_batch_slp_run: launches of different scenarios merged, scenario index in the grid)" % (per_gpu, n_slots, hb.groups),
                          "inputs": "scenario NLPs built on the host before the timed region; bounds uploads and solves inside it"},
               "batch_stats": stats,
               "lockstep": {"slots": n_slots, "groups": hb.groups, "rounds": d["rounds"], "recorded_launches": d["ops"], "launches": d["launches"],
                            "merge_ratio": d["ops"] / max(d["launches"], 1), "argument_table_MB": d["blob_bytes"] / 1e6,
                            "emit_ms": d["emit_ms"], "device_wait_ms": d["wait_ms"], "solver_host_ms": d["host_ms"], "wall_ms": d["wall_ms"]},
               "lp_outcomes": {"paths": hist, "lps": int(sum(r.lp_solves for r in runs)), "non_canonical_answers": int(sum(r.paths[9] + r.paths[10] for r in runs)),
                               "unpolished": int(sum(r.paths[5] for r in runs)), "cold_basis_selections": int(sum(r.ns_cold for r in runs)),
                               "ipm_iterations_per_lp": sum(r.ipm_iters for r in runs) / max(1, sum(r.lp_solves for r in runs))},
               "host": {"cpu_quota": _host_cpu_quota(), "blas_threads": HOST_THREADS, "cgroup_throttled_s": round(thr, 3),
                        "rank_seconds": [round(v, 3) for v in rank_s], "rank_throttled_s": [round(v, 3) for v in rank_thr]}}
        # roofline of the dominant kernel family as the batch runs it (rank 0's GPU): the dataflow panel kernels, HIP events around every MERGED
        # launch on its group's stream (include/asm_hip.h: asm_batch_stats.panel_*); algorithmic flops = what the slots' factorisations account
        if d.get("panel_launches", 0) > 0 and d.get("panel_ms", 0.0) > 0.0:
            tf = d["panel_flops"] / (d["panel_ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "k_chol_panel_solo + k_chol_panel_inv (dataflow panel kernels) as merged launches of the lockstep batch: "
                                                          "one launch factors the same inner panel of several scenarios - a dependent chain of 64-wide steps, latency-bound",
                               "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                               "avg_launch_ms": d["panel_ms"] / d["panel_launches"], "launches": int(d["panel_launches"]),
                               "scenario_launches_merged": int(d["panel_ops"]), "scenarios_per_launch": d["panel_ops"] / d["panel_launches"],
                               "share_of_stream_time": d["panel_ms"] / max(1e-9, 1e3 * mine * hb.groups),
                               "algorithmic_bytes_per_launch": d["panel_bytes"] / d["panel_launches"],
                               "note": "rank 0; the groups' streams run side by side, so the family's share is taken of groups x wall time; traffic: no PMC pass of the batch is committed"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = c5_cpu_baseline(base, lo, args, runs)
        print(json.dumps(out))
    hb.close()
    if dist is not None:
        dist.destroy_process_group()


def c5_cpu_baseline(base, sidx, args, runs):
    """CPU comparator of the scenario batch on rank 0's host cores: SciPy's HiGHS dual simplex (sparse, cold start, 1 thread) on the sequence of
    sub-LPs of ONE scenario (its iterates are recorded by a per-handle run of the same solver outside the timed region), for at most
    --cpu-seconds; solves/s = LPs/s divided by the batch's mean LP count per scenario solve."""
    try:
        import activesetmethods_amd as A
        from activesetmethods_amd import acopf
        case = acopf.scenario_case(base, sidx)
        prd = acopf.function_model(case).to_problem("case300-sized scenario %d" % sidx)
        slp = A.optimize(A.Model.from_problem(prd, A.Parameters(algorithm=args.algorithm, max_iter=args.max_iter, device_eval=True)))
        slp.optimizer.close()
        hs = highs_sequence(acopf.acopf_problem(case, "case300-sized scenario"), slp.trace, args.cpu_seconds)
        if "error" in hs:
            return hs
        lps_per_solve = sum(r.lp_solves for r in runs) / max(len(runs), 1)
        return dict(value=hs["value"] / lps_per_solve, unit="solves/s", cores=1, kind="port",
                    sample="HiGHS dual simplex (scipy.optimize.linprog, sparse, cold start, 1 thread) on the first %d sub-LPs of scenario %d's SLP run "
                           "(%.1f s, %.2f LPs/s), divided by the batch's mean of %.1f LPs per scenario solve; LP solve time only - evaluation, "
                           "assembly and the line search are not charged to the CPU" % (hs["lp_solved"], sidx, hs["seconds"], hs["value"], lps_per_solve),
                    highs_sequence=hs)
    except Exception as e:
        return dict(error=repr(e))


def run_batch_pool(args, rank, world, local_rank, dist, torch, base, problems, total, per_gpu):
    """Round-3 path of workload c5 (--batch-mode pool): `--concurrency` host threads, each with one persistent handle = one HIP stream and
    the Python SLP driver."""
    import activesetmethods_amd as A
    from activesetmethods_amd import batch
    import threading
    tls = threading.local()                 # one handle per worker thread of the stream pool, kept for the whole batch

    def factory(d, r, c):                   # scenarios differ in constraint bounds only (the loads): same pattern, same LP skeleton
        opt = getattr(tls, "opt", None)
        if opt is None:
            opt = tls.opt = A.HipSubOptimizer(d, r, c, device=local_rank)
        else:
            opt.set_bounds(d)               # keeps J, Ah, S, the assembly plan and the evaluator's function store in HBM
        return opt

    def make_model(sidx):
        return A.Model.from_problem(problems[sidx], A.Parameters(algorithm=args.algorithm, max_iter=args.max_iter, external_optimizer=factory,
                                                                 device_eval=not args.host_eval))

    models = {s: make_model(s) for s in problems}
    _freeze_heap()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    thr0 = _cpu_throttled_s()
    t0 = time.perf_counter()
    slps, stats = batch.solve_batch(models.__getitem__, total, rank, world, run=lambda m: A.optimize(m), reduce_device=args.reduce_device,
                                    concurrency=args.concurrency)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=args.reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        out = {"metric": "batch-NLP solves/sec", "value": total / elapsed, "unit": "solves/s", "n_gpus": world, "steps": per_gpu,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / per_gpu, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": WORKLOADS["c5"]["desc"], "algorithm": args.algorithm, "max_iter": args.max_iter,
                          "scenarios_total": total,
                          "parallelism": "scenarios block-partitioned, %d per GPU, %d at a time per GPU (stream pool, round-3 path)" % (per_gpu, args.concurrency)},
               "batch_stats": stats,
               "host": {"cpu_quota": _host_cpu_quota(), "blas_threads": HOST_THREADS, "cgroup_throttled_s": round(_cpu_throttled_s() - thr0, 3)}}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def _freeze_heap():
    """The imports (torch above all) leave ~170 000 long-lived objects on the collector's lists; every full collection the SLP
    driver's per-step allocations trigger walks them all.  They are moved to the permanent generation before the timed region."""
    import gc
    gc.collect()
    gc.freeze()


def make_problem(name, device_eval=True):
    """The workload as the reference receives it: a model of affine / quadratic functions + an NLP block in the MOI wrapper's
    lists (activesetmethods_amd/moi_evaluator.py); f, grad f, g and the Jacobian values are evaluated by the device kernels and the
    Jacobian values never leave HBM.  Second return value: the same NLP with vectorised NumPy callbacks (rows in another order) -
    what --host-eval runs and what the CPU baselines evaluate with (the FunctionModel's own host evaluator is a term-by-term
    Python loop, fine as a checker, unfair as a baseline)."""
    from activesetmethods_amd import problems, acopf
    if name in ("c3", "c4", "c4fr"):
        wl = WORKLOADS[name]
        case = acopf.synthetic_case(wl["case"], 1, wl["load_scale"])
        label = "%s-sized, load scale %g" % (wl["case"], wl["load_scale"])
        host = acopf.acopf_problem(case, label)
        return (acopf.function_model(case).to_problem(label) if device_eval else host), host
    n, m = (1000, 500) if name == "c2" else (200, 100)
    host = problems.synthetic_dense_nlp(n, m)
    return (problems.synthetic_dense_function_model(n, m).to_problem(host.name) if device_eval else host), host


def run_steps(pr, algorithm, device, n_steps, state):
    """Advance the SLP run by `n_steps` LP solves (restarting from x0 if the run terminates)."""
    import activesetmethods_amd as A
    done = 0
    while done < n_steps:
        if state.get("slp") is None:
            par = A.Parameters(algorithm=algorithm, max_iter=10 ** 9, device_eval=state.get("device_eval", True),
                               external_optimizer=lambda d, r, c: A.HipSubOptimizer(d, r, c, device=device))
            mdl = A.Model.from_problem(pr, par)
            slp = A.SlpTR(mdl) if algorithm == "Trust Region" else A.SlpLS(mdl)
            state["slp"], state["resume"] = slp, False
            if state.get("opt") is not None:            # keep one handle (its HBM buffers and the evaluator's function store) per rank
                slp.optimizer = state["opt"]
        slp = state["slp"]
        before = slp.lp_solves
        ntrace = len(slp.trace)
        target = before + (n_steps - done)
        slp.run(max_lp_solves=target, resume=state["resume"])
        state.setdefault("trace_all", []).extend(slp.trace[ntrace:])
        state["resume"] = True
        state["opt"] = slp.optimizer
        done += slp.lp_solves - before
        if slp.lp_solves < target:                      # the SLP run terminated: restart from x0
            state["restarts"] = state.get("restarts", 0) + 1
            state.setdefault("terminations", []).append(int(slp.ret))
            state["slp"] = None
    return done


def _cores():
    try:
        import threadpoolctl
        return int(min(_host_cpu_quota(), max((p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()), default=1)))
    except Exception:
        return HOST_THREADS


def highs_sequence(pr, timed, budget_s):
    """Independent third-party CPU LP code on the SEQUENCE of sub-LPs the timed GPU steps solved: for every
    timed step, in order, the LP is rebuilt at that step's iterate (oracle/sparse_lp.py: the sparse statement of
    subproblem.jl:229-484, normal or restoration phase as the step had it) and solved by SciPy's HiGHS dual simplex, cold
    start, one thread, until the time budget is spent.  This is the natural CPU method for these LPs (a sparse simplex, as
    the reference's GLPK) - but not GLPK itself (absent here) and without GLPK's retained-basis warm start (linprog has none)."""
    try:
        import numpy as np
        from oracle import sparse_lp
        t_tot, n_done, n_fr, mism, its = 0.0, 0, 0, 0, 0
        for rec in timed:
            if t_tot >= budget_s:
                break
            x = np.asarray(rec["x"], float)
            lp = sparse_lp.build(pr.n, pr.m, pr.j_row, pr.j_col, pr.eval_jac_g(x, np.zeros(pr.nnz)), pr.eval_grad_f(x, np.zeros(pr.n)),
                                 pr.eval_g(x, np.zeros(pr.m)), pr.g_L, pr.g_U, pr.x_L, pr.x_U, x, rec["delta"], rec["fr"])
            st, obj, p, dt, nit = sparse_lp.solve_highs(lp)
            t_tot += dt
            n_done += 1
            n_fr += 1 if rec["fr"] else 0
            its += nit
            mism += 1 if st != rec["status"] else 0
        if n_done == 0:
            return dict(error="no timed step")
        return dict(value=n_done / t_tot, unit="iter/s", cores=1, lp_solved=n_done, restoration_lps=n_fr, seconds=t_tot,
                    simplex_iterations=its, status_mismatches_vs_gpu=mism,
                    note="scipy.optimize.linprog(method='highs-ds') on the sub-LPs of the first %d timed steps (their iterates, radius and phase), "
                         "sparse, cold start, 1 thread; LP solve time only" % n_done)
    except Exception as e:                                   # SciPy missing or HiGHS failure: report, do not fail the bench
        return dict(error=repr(e))


def check_non_canonical(pr, recs, tol=1e-6):
    """HiGHS (independent LP code) on the LPs whose GPU answer is not the canonical pair: optimal value and feasibility of the GPU's step."""
    try:
        import numpy as np
        from oracle import sparse_lp
        worst, fails = 0.0, 0
        for rec in recs:
            x = np.asarray(rec["x"], float)
            lp = sparse_lp.build(pr.n, pr.m, pr.j_row, pr.j_col, pr.eval_jac_g(x, np.zeros(pr.nnz)), pr.eval_grad_f(x, np.zeros(pr.n)),
                                 pr.eval_g(x, np.zeros(pr.m)), pr.g_L, pr.g_U, pr.x_L, pr.x_U, x, rec["delta"], rec["fr"])
            st, obj, _, _, _ = sparse_lp.solve_highs(lp)
            fixed = dict(lp)
            b = lp["bounds"].copy()
            b[:pr.n, 0] = b[:pr.n, 1] = np.asarray(rec["p"], float)
            fixed["bounds"] = b
            st2, obj2, _, _, _ = sparse_lp.solve_highs(fixed)
            if st != 1 or st2 != 1:
                fails += 1
                continue
            gap = abs(obj2 - obj) / max(1.0, abs(obj))
            worst = max(worst, gap)
            fails += 1 if gap > tol else 0
        return dict(checked=len(recs), failures=fails, max_rel_objective_gap=worst, tolerance=tol,
                    note="scipy HiGHS on the same LP: optimal value vs the LP's value at the GPU's step (step fixed, slacks re-optimised); an infeasible fixed LP counts as failure")
    except Exception as e:
        return dict(error=repr(e))


def cpu_baseline(pr, algorithm, budget_s, mix, timed, ns_J=None):
    """Oracle (NumPy restatement of the same path) on a bounded sample of the same workload.
    Small NLPs: whole SLP iterations from x0.  Large NLPs (one LP would take minutes on the host): single
    interior-point iterations of the oracle are timed in both forms of the Newton system and priced with the mix of
    iterations (`mix`: row-form, column-form, active-set solves) the timed GPU steps actually took."""
    import numpy as np
    from oracle import slp as O
    from oracle import lp_solver as L
    from oracle.subproblem import QpData, QpModel, compute_jacobian_matrix
    cores = _cores()
    if pr.n * pr.m <= 2_000_000:
        steps, t_total, k = 0, 0.0, 1
        while True:
            m = O.Model(pr.n, pr.m, pr.x_L, pr.x_U, pr.g_L, pr.g_U, pr.j_str, pr.eval_f, pr.eval_g, pr.eval_grad_f, pr.eval_jac_g,
                        O.Parameters(algorithm=algorithm, max_iter=k))
            m.x[:] = pr.x0
            t0 = time.perf_counter()
            s = O.optimize(m)
            dt = time.perf_counter() - t0
            steps, t_total = s.lp_solves, dt
            if dt >= 0.5 * budget_s or k >= 64:
                break
            k = max(k + 1, int(k * min(4.0, 0.8 * budget_s / max(dt, 1e-3))))
        x = pr.x0.copy()
        A, st = compute_jacobian_matrix(pr.m, pr.n, pr.j_row - 1, pr.j_col - 1, pr.eval_jac_g(x, np.zeros(pr.nnz)))
        qp = QpModel(QpData(pr.eval_grad_f(x, np.zeros(pr.n)), pr.eval_f(x), A, pr.eval_g(x, np.zeros(pr.m)), pr.g_L, pr.g_U, pr.x_L, pr.x_U, st),
                     pr.j_row, pr.j_col)
        return dict(value=steps / t_total, unit="iter/s", cores=cores, kind="port",
                    sample="oracle (NumPy restatement of the same dense algorithm) SLP (%s) from x0, first %d SLP iterations of the same NLP, %.1f s"
                           % (algorithm, steps, t_total),
                    highs_sequence=highs_sequence(pr, timed, budget_s))
    # large NLPs: ONE sub-LP of the timed sequence (the last timed step: its iterate, radius and phase) solved by the oracle end to
    # end - assembly, formulation, scaling, the LP solve with the same algorithm the GPU runs.  Normal-phase LPs: the columns of
    # the null-space basis are the ones the GPU run retained (asm_sublp_ns_basis), so the oracle skips its own from-scratch
    # selection exactly as every LP after the first does; restoration LPs (dense row / column forms, minutes per LP on the host): the
    # round-2 pricing - single interior-point iterations in both forms, priced with the GPU run's own mix.
    rec = timed[-1] if timed else None
    if rec is not None and not rec["fr"]:
        x = np.asarray(rec["x"], float)
        t0 = time.perf_counter()
        dE = pr.eval_jac_g(x, np.zeros(pr.nnz))
        A, st = compute_jacobian_matrix(pr.m, pr.n, pr.j_row - 1, pr.j_col - 1, dE)
        qp = QpModel(QpData(pr.eval_grad_f(x, np.zeros(pr.n)), pr.eval_f(x), A, pr.eval_g(x, np.zeros(pr.m)), pr.g_L, pr.g_U, pr.x_L, pr.x_U, st),
                     pr.j_row, pr.j_col)
        if ns_J is not None and len(ns_J):
            qp.hint[False]['ns_J'] = np.asarray(ns_J, np.int64)
        out = qp.sub_optimize(x, rec["delta"], False)
        dt = time.perf_counter() - t0
        stt = out[6]['stats']
        return dict(value=1.0 / dt, unit="iter/s", cores=cores, kind="port",
                    sample="oracle (NumPy restatement of the same algorithm incl. the null-space form; scipy.sparse products, LAPACK Cholesky) on ONE sub-LP of the timed "
                           "sequence, end to end (assembly + formulation + scaling + LP solve): %.1f s, status %d, path %s, %d interior-point "
                           "iterations (%d in null-space form, dimension %d), basis columns %s"
                           % (dt, out[5], stt.get('path'), stt.get('ipm_iters', 0), stt.get('ns_iters', 0), len(ns_J) if ns_J is not None else 0,
                              "taken from the GPU run" if ns_J is not None and len(ns_J) and not stt.get('ns_cold', 1) else "selected from scratch"),
                    highs_sequence=highs_sequence(pr, timed, budget_s))
    x = pr.x0.copy()
    t0 = time.perf_counter()
    dE = pr.eval_jac_g(x, np.zeros(pr.nnz))
    A, st = compute_jacobian_matrix(pr.m, pr.n, pr.j_row - 1, pr.j_col - 1, dE)
    qp = QpModel(QpData(pr.eval_grad_f(x, np.zeros(pr.n)), pr.eval_f(x), A, pr.eval_g(x, np.zeros(pr.m)), pr.g_L, pr.g_U, pr.x_L, pr.x_U, st),
                 pr.j_row, pr.j_col)
    delta = 1000.0 if algorithm == "Line Search" else 0.4
    slp_fr, _, _, _ = L.scale_lp(qp.build_lp(x, delta, True))
    t_setup = time.perf_counter() - t0

    def time_iterations(scaled_lp, budget, cap, col):
        ip = L.IPM(scaled_lp)
        ip.ns_ok = False
        if not col:
            ip.col_ok = False
        its, t_ipm = 0, 0.0
        while t_ipm < budget and its < cap:
            t1 = time.perf_counter()
            ip.run(0.0, 1)
            t_ipm += time.perf_counter() - t1
            its += 1
        return t_ipm / its, its, ip

    t_row, n_row, _ = time_iterations(slp_fr, 0.6 * budget_s, 3, False)
    t_col, n_col = t_row, 0
    if mix["col_iters"] > 0:
        t_col, n_col, ipc = time_iterations(slp_fr, 0.4 * budget_s, 3, True)
        if not ipc.col_ok:
            t_col = t_row
    eqp_cost = t_row * 0.2                      # an active-set factorisation has ~0.58 M rows: 0.58^3
    step_s = t_setup + (mix["row_iters"] * t_row + mix["col_iters"] * t_col + mix["eqp"] * eqp_cost) / max(mix["steps"], 1)
    return dict(value=1.0 / step_s, unit="iter/s", cores=cores, kind="port",
                sample="oracle (NumPy restatement of the same dense algorithm) on the restoration LP at x0: assembly+formulation+scaling (%.1f s), %d row-form "
                       "interior-point iterations (%.1f s each) and %d column-form iterations (%.1f s each), priced with the GPU run's own mix per SLP "
                       "step: %.1f row-form + %.1f column-form iterations + %.1f active-set factorisations (0.2 row-form iterations each)"
                       % (t_setup, n_row, t_row, n_col, t_col, mix["row_iters"] / max(mix["steps"], 1), mix["col_iters"] / max(mix["steps"], 1),
                          mix["eqp"] / max(mix["steps"], 1)),
                highs_sequence=highs_sequence(pr, timed, budget_s))


def launch_ranks(n, argv, script=None):
    """Start `n` ranks of `script` (default: this file) as child processes, one per GPU, with the rendezvous variables of
    torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT).  Called before anything
    touches the GPU (a process that has initialised HIP must not fork/exec workers).  Returns the first non-zero exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:               # a failed rank would leave the others waiting at the next barrier
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c4 on one GPU (the configuration the metric is quoted on), c5 (scenario batch) on several")
    ap.add_argument("--algorithm", default=None)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--scenarios-per-gpu", type=int, default=None, help="workload c5: scenarios per GPU (alias of --steps)")
    ap.add_argument("--max-iter", type=int, default=100, help="workload c5: SLP iteration cap per scenario")
    ap.add_argument("--slots", type=int, default=64, help="workload c5: scenarios advancing in lockstep per GPU (slots of the asm_batch)")
    ap.add_argument("--groups", type=int, default=None, help="workload c5: groups of slots (one stream + one host thread each; library default: 2 from 16 slots on)")
    ap.add_argument("--dynamic", action="store_true", help="workload c5: ranks claim chunks of scenarios from a shared counter instead of the static block partition")
    ap.add_argument("--batch-mode", default="lockstep", choices=["lockstep", "pool"], help="workload c5: lockstep batch (default) or the round-3 stream pool")
    ap.add_argument("--concurrency", type=int, default=2, help="workload c5, --batch-mode pool: scenarios in flight per GPU (one handle / HIP stream each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-eval", action="store_true", help="evaluate f, grad f, g, Jacobian values with the NumPy callbacks instead of the device kernels")
    ap.add_argument("--kernel-breakdown", action="store_true",
                    help="time every kernel family with HIP events (adds kernels_ms; default: only the dominant kernel, k_syrk, is timed)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.workload is None:
        args.workload = "c4" if world == 1 else "c5"
    wl = WORKLOADS[args.workload]
    args.steps = wl["steps"] if args.steps is None else args.steps
    args.warmup = wl["warmup"] if args.warmup is None else args.warmup
    args.algorithm = wl["algorithm"] if args.algorithm is None else args.algorithm

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HIP-event timing inside the library: 1 = the dominant kernel only (two events per k_syrk launch, on the stream it is
    # launched on), 2 = every kernel family (perturbs the latency-bound small workloads)
    os.environ.setdefault("ASM_HIP_TIMING", "2" if args.kernel_breakdown else "1")
    if args.workload == "c5":
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # one hardware queue per concurrent scenario stream (ROCm default: 4)
        import torch as _t
        nvis = _t.cuda.device_count()
        if nvis and world > nvis:
            # rehearsal of N ranks on fewer GPUs: the ranks that share a device share its compute units - the all-resident panel kernels of
            # all of them must fit it together (asm_create reads ASM_PANEL_WGS)
            os.environ.setdefault("ASM_PANEL_WGS", str(max(16, 480 // ((world + nvis - 1) // nvis))))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    rehearsal = ndev < world            # more ranks than GPUs (1-GPU box): ranks share devices, statistics go over gloo
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))      # RCCL over xGMI
    args.reduce_device = None if dist is None else ("cpu" if rehearsal else "cuda")

    if args.workload == "c5":
        if args.scenarios_per_gpu is not None:
            args.steps = args.scenarios_per_gpu
        return run_batch(args, rank, world, local_rank, dist, torch)
    pr, pr_host = make_problem(args.workload, not args.host_eval)
    state = {"device_eval": not args.host_eval}
    run_steps(pr, args.algorithm, local_rank, args.warmup, state)
    opt = state["opt"]
    opt.kernel_stats(reset=True)
    _freeze_heap()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    done = run_steps(pr, args.algorithm, local_rank, args.steps, state)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    assert done == args.steps
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=args.reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ks = opt.kernel_stats()
    trace = list(state.get("trace_all", []))        # (the all-families pass below appends to the live list)
    # ---- roofline.  Every launch of the two MFMA kernel families is timed with HIP events on the stream it runs on:
    #   panel_kernel   k_chol_panel / _solo / _inv / _band: the 64-wide steps of the Cholesky factorisations (a dependent chain: latency-bound)
    #   syrk_kernel    k_syrk_upd + k_syrk<T>: rank-K updates of the factorisations and the Schur / reduced-matrix builds
    # `roofline` describes the family with the larger share of the step; both are listed under `families`.
    fams = {}
    for key, label in (("panel_kernel", "k_chol_panel / _solo / _inv / _band (dataflow panel kernels: factor + inverse of the 64 x 64 diagonal block, panel solve, rank-64 "
                                        "update per step; _inv also forms the explicit inverse of a one-wide-block factor and _band the whole trailing update of a "
                                        "banded factor by helper workgroups; a dependent chain of 64-wide steps - latency-bound, the matrix pipe idles between them)"),
                       ("syrk_kernel", "k_syrk_upd + k_syrk<T> (f64 MFMA rank-K kernels: Cholesky updates, Schur and reduced-matrix builds)")):
        d = ks[key]
        a = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        fams[key] = dict(kernel=label, achieved=a, frac=a / FP64_MFMA_PEAK_TFLOPS, avg_launch_ms=d["ms"] / max(d["calls"], 1), launches=d["calls"],
                         share_of_step_time=d["ms"] / (1e3 * elapsed) if elapsed > 0 else None, algorithmic_bytes_per_launch=d["bytes"] / max(d["calls"], 1))
    dom = max(fams, key=lambda k: fams[k]["share_of_step_time"] or 0.0)
    roof = dict(bound="mfma", kernel=fams[dom]["kernel"], achieved=fams[dom]["achieved"], peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=fams[dom]["frac"],
                traffic=None, avg_launch_ms=fams[dom]["avg_launch_ms"], launches=fams[dom]["launches"], share_of_step_time=fams[dom]["share_of_step_time"],
                algorithmic_bytes_per_launch=fams[dom]["algorithmic_bytes_per_launch"], families=fams)
    nfact = ks["chol"]["calls"]
    # a second, untimed pass with every kernel family timed (perturbs the latency-bound chains, hence not the measured one): the whole
    # step's algorithmic work and the HBM regime of SURVEY.md 8(d) (matrix-vector products, substitutions, assembly, scaling)
    opt.kernel_timing(2)
    opt.kernel_stats(reset=True)
    n2 = max(3, min(10, args.steps))
    t2 = time.perf_counter()
    run_steps(pr, args.algorithm, local_rank, n2, state)
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t2
    k2 = opt.kernel_stats()
    opt.kernel_timing(1)
    step_flops = sum(k2[k]["flops"] for k in ("syrk", "chol", "trsv", "gemv")) / n2
    step_bytes = sum(k2[k]["bytes"] for k in ("assemble", "scale", "gemv", "trsv", "syrk", "chol")) / n2
    roof["step"] = dict(flops=step_flops, bytes=step_bytes, tflops=step_flops / (1e-3 * 1e3 * elapsed / args.steps) / 1e12,
                        frac=step_flops / (elapsed / args.steps) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                        note="algorithmic flops / bytes of one step (Schur and reduced-matrix builds + factorisations + multi-right-hand-side and vector "
                             "substitutions + matrix-vector products; assembly and scaling bytes) from the all-families pass, divided by the timed ms_per_step")
    hb_ms = sum(k2[k]["ms"] for k in ("assemble", "scale", "gemv"))
    hb_bytes = sum(k2[k]["bytes"] for k in ("assemble", "scale", "gemv"))
    roof["hbm"] = dict(bound="hbm", kernels="k_assemble, scaling passes, k_spmv_* / k_gemv_* (matrix-vector products)", achieved=hb_bytes / (hb_ms * 1e-3) / 1e9 if hb_ms > 0 else 0.0,
                       peak=HBM_PEAK_GBS, unit="GB/s", frac=(hb_bytes / (hb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if hb_ms > 0 else 0.0,
                       share_of_step_time=(hb_ms / n2) / (1e3 * dt2 / n2) if dt2 > 0 else None,
                       note="HBM regime of SURVEY.md 8(d): algorithmic bytes / HIP-event time of these families in the all-families pass; at these sizes "
                            "(13 k - 64 k non-zeros) the launches are latency-bound, not bandwidth-bound")
    roof["all_families_pass"] = dict(steps=n2, ms_per_step=1e3 * dt2 / n2, kernels_ms_per_step={k: round(v["ms"] / n2, 3) for k, v in k2.items()})
    # HBM bytes per launch of the dominant family from the committed PMC passes (cannot be collected inside bench)
    for tag in ("r04", "r03", "r02"):
        tpath = os.path.join(ROOT, "profiles", "%s_%s_pmc_traffic.json" % (tag, args.workload))
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            famj = tj.get("families", {})
            key = dom if dom != "syrk_kernel" or args.workload in ("c2", "c4fr") else "syrk_kernel_without_upd"      # k_syrk_upd: only in the cold first LP of c3 / c4
            if key in famj:
                roof["traffic"] = famj[key]["traffic_bytes_per_launch"]
            elif dom == "syrk_kernel":
                roof["traffic"] = tj.get("traffic_bytes_per_launch")
            roof["traffic_source"] = "profiles/" + os.path.basename(tpath)
            break
    if rank == 0:
        w = WORKLOADS[args.workload]
        out = {
            "metric": "SLP iterations/sec", "value": world * args.steps / elapsed, "unit": "iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["desc"], "algorithm": args.algorithm, "n": pr.n, "m": pr.m, "nnz": pr.nnz,
                       "parallelism": "replicas x%d" % world, "restarts": state.get("restarts", 0),
                       "evaluation": "host NumPy callbacks, dE uploaded per step" if args.host_eval else "device kernels (dE stays in HBM)",
                       "factorisations_per_step": nfact / max(args.steps, 1)},
            "roofline": roof,
        }
        if args.kernel_breakdown:
            out["kernels_ms"] = {k: round(v["ms"], 3) for k, v in ks.items()}
        timed = trace[-args.steps:] if trace else []
        names = {0: "warm", 1: "ipm0+ln", 2: "ipm1+ln", 3: "ipm2+ln", 4: "ipm+face", 5: "ipm-unpolished", 6: "ipm-infeasible",
                 7: "phase1-infeasible", 8: "ipm~+ln", 9: "ipm+ref", 10: "ipm-conv"}
        hist = {}
        for r in timed:
            key = names.get(r["stats"]["path"], str(r["stats"]["path"])) + ("/fr" if r["fr"] else "")
            hist[key] = hist.get(key, 0) + 1
        def _mean(v):
            return (sum(v) / len(v)) if v else None
        nm = [r for r in timed if not r["fr"]]
        frs = [r for r in timed if r["fr"]]
        out["phases"] = {      # restoration solves reported separately (LP wall time inside the library, per LP)
            "normal_lps": len(nm), "restoration_lps": len(frs),
            "lp_ms_normal": _mean([r["stats"]["wall_ms"] for r in nm]), "lp_ms_restoration": _mean([r["stats"]["wall_ms"] for r in frs]),
            "ipm_iterations_per_lp": _mean([r["stats"]["ipm_iters"] for r in timed]),
            "null_space_iterations_per_lp": _mean([r["stats"].get("ns_iters", 0) for r in timed]),
            "null_space_dimension": max([r["stats"].get("ns_dim", 0) for r in timed], default=0),
            "factorisations_per_lp_all_sizes": _mean([r["stats"]["nfact"] for r in timed])}
        out["lp_outcomes"] = {"paths": hist, "unpolished": sum(1 for r in timed if r["stats"]["polished"] != 1),
                              "non_canonical_answers": sum(1 for r in timed if r["stats"]["path"] in (9, 10)),     # 'ipm+ref': projection of the iterate; 'ipm-conv': the converged iterate itself
                              "status_other": sum(1 for r in timed if r["status"] not in (1, 2)),
                              "restoration_lps": sum(1 for r in timed if r["fr"]),
                              "slp_status_last": int(state["slp"].ret) if state.get("slp") is not None else None,
                              "slp_terminations": state.get("terminations", [])}
        # every non-canonical answer of the timed steps ('ipm+ref': projection of the iterate, 'ipm-conv': the converged iterate) against an
        # independent LP code: HiGHS optimal value of the same LP vs the value of the LP at the GPU's step (its slacks re-optimised by HiGHS
        # with p fixed: restoration LPs), and feasibility of that step.  A failure is reported, not hidden.
        nc = [r for r in timed if r["stats"]["path"] in (9, 10) and r["status"] == 1][:5]
        if nc:
            out["lp_outcomes"]["non_canonical_check"] = check_non_canonical(pr_host, nc)
            if out["lp_outcomes"]["non_canonical_check"].get("failures", 0) > 0:
                out["parity_failures"] = out["lp_outcomes"]["non_canonical_check"]["failures"]
        if not args.no_cpu_baseline and world == 1:
            mix = dict(steps=args.steps, col_iters=sum(r["stats"].get("col_iters", 0) for r in timed),
                       row_iters=sum(r["stats"]["ipm_iters"] - r["stats"].get("col_iters", 0) for r in timed),
                       eqp=sum(max(r["stats"]["nfact"] - r["stats"]["ipm_iters"], 0) for r in timed))   # active-set FACTORISATIONS
            if not timed:
                mix = dict(steps=args.steps, col_iters=0, row_iters=nfact, eqp=0)
            out["cpu_baseline"] = cpu_baseline(pr_host, args.algorithm, args.cpu_seconds, mix, timed, opt.ns_basis())
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
