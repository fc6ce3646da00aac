"""CPU suite for everything around the HIP kernels: the C ABI library loads and exports exactly the
symbols include/asm_hip.h declares (no compute call without a GPU), the host-side SLP drivers of the
package reproduce the oracle's callers when driven through the `external_optimizer` plug-in slot, and
the scenario-batch statistics reduce correctly over a world_size-2 gloo group."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "asm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(asm_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from activesetmethods_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "libasmhip.so does not export %s" % s
    assert sorted(_lib.PROTOTYPES) == syms          # the Python binding covers the whole header


def test_missing_gpu_fails_loudly():
    """No silent fallback: without a HIP device the sub-optimizer cannot be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from activesetmethods_amd import HipSubOptimizer, QpData, AsmHipError
    d = QpData(np.ones(2), 0.0, np.ones(2), np.zeros(1), np.zeros(1), np.zeros(1), -np.ones(2), np.ones(2))
    with pytest.raises(AsmHipError):
        HipSubOptimizer(d, [1, 1], [1, 2])


class _OracleBackedSubOptimizer:
    """Stand-in for the HIP plug-in so that the package's SLP drivers can be exercised without a GPU.
    (Test-only: the shipped default factory is HipSubOptimizer.)"""

    def __init__(self, data, j_row, j_col):
        from oracle.subproblem import QpModel
        self.j_row, self.j_col = np.asarray(j_row), np.asarray(j_col)
        self.data = data
        self.qp = None
        self._QpModel = QpModel
        self._info = None

    def upload(self, dE, df, f, E, x_k):
        from oracle.subproblem import QpData, compute_jacobian_matrix
        d = self.data
        m, n = len(d.c_lb), len(d.v_lb)
        A, st = compute_jacobian_matrix(m, n, self.j_row - 1, self.j_col - 1, np.array(dE))
        self.A = A
        od = QpData(np.array(df), f, A, np.array(E), d.c_lb, d.c_ub, d.v_lb, d.v_ub, st)
        if self.qp is None:
            self.qp = self._QpModel(od, self.j_row, self.j_col)
        self.qp.data = od
        self.x_k = np.array(x_k)

    def solve_resident(self, Delta, feasibility=False):
        out = self.qp.sub_optimize(self.x_k, Delta, feasibility)
        self._info = out[6]
        return out[:6]

    def last_stats(self):
        s = dict(self._info["stats"])
        s["path"] = 0 if s["path"] == "warm" else 1
        return s

    def active_set(self):
        return self._info["sets"]

    def kt_residuals(self, df, lam, mU, mL):
        from oracle.slp import KT_residuals
        return KT_residuals(df, lam, mU, mL, self.A)

    def jac_row_norms(self):
        return np.linalg.norm(self.A, axis=1)


@pytest.mark.parametrize("alg", ["Line Search", "Trust Region"])
def test_host_drivers_follow_the_reference_callers(alg):
    import activesetmethods_amd as A
    from oracle import slp as O
    for pr, iters in ((A.problems.toy_problem(), 1000), (A.problems.synthetic_dense_nlp(40, 20), 6)):
        mh = A.Model.from_problem(pr, A.Parameters(algorithm=alg, max_iter=iters, external_optimizer=_OracleBackedSubOptimizer))
        sh = A.optimize(mh)
        mo = O.Model(pr.n, pr.m, pr.x_L, pr.x_U, pr.g_L, pr.g_U, pr.j_str, pr.eval_f, pr.eval_g, pr.eval_grad_f, pr.eval_jac_g,
                     O.Parameters(algorithm=alg, max_iter=iters))
        mo.x[:] = pr.x0
        so = O.optimize(mo)
        assert mh.status == mo.status and sh.iter == so.iter and sh.lp_solves == so.lp_solves
        assert np.abs(mh.x - mo.x).max() < 1e-10
        assert np.abs(mh.mult_g - mo.mult_g).max() < 1e-9
        assert [r["fr"] for r in sh.trace] == [r["fr"] for r in so.trace]


def test_parameters_defaults_are_the_reference_contract():
    import activesetmethods_amd as A
    p = A.Parameters()
    assert (p.tol_direction, p.tol_residual, p.tol_infeas, p.max_iter) == (1e-6, 0.01, 0.01, 1000)    # src/parameters.jl:17-20
    assert (p.eta, p.tau, p.min_alpha, p.tr_size) == (0.4, 0.9, 1e-6, 0.4)                           # :25-28
    assert p.algorithm == "Line Search" and p.method == "SLP"
    A.set_parameter(p, "max_iter", 7)
    assert A.get_parameter(p, "max_iter") == 7
    assert A.ApplicationReturnStatus[0] == "Solve_Succeeded" and A.ApplicationReturnStatus[-12] == "Invalid_Option"


def test_partition_covers_all_scenarios():
    from activesetmethods_amd.batch import partition
    for n, w in ((512, 8), (10, 3), (2, 4)):
        spans = [partition(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert partition(512, 8, 3) == (192, 256)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    import activesetmethods_amd as A
    from activesetmethods_amd.batch import solve_batch
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)

    def make_model(s):                      # scenario s: toy NLP started from a scenario-dependent point
        pr = A.problems.toy_problem()
        mdl = A.Model.from_problem(pr, A.Parameters(algorithm="Line Search", external_optimizer=_OracleBackedSubOptimizer))
        mdl.x[:] = [0.1 * s, 0.0]
        return mdl

    slps, stats = solve_batch(make_model, 5, rank, world)
    q.put((rank, len(slps), stats))
    dist.destroy_process_group()


def test_batch_statistics_all_reduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort()
    assert [r[1] for r in res] == [3, 2]                         # block partition of 5 scenarios
    s0, s1 = res[0][2], res[1][2]
    assert s0 == s1                                              # every rank holds the reduced statistics
    assert s0["scenarios"] == 5 and s0["converged"] == 5
    assert s0["lp_solves"] >= s0["iterations"] - 5 and s0["wall_s"] > 0


def test_bench_launcher_starts_one_rank_per_gpu(tmp_path):
    """`bench.py --gpus N` (N > 1, no WORLD_SIZE) starts N ranks itself with the torch.distributed.run rendezvous
    variables; here the ranks are a stand-in script that joins a gloo group and all-reduces its rank."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys, json, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(os.environ['RANK'])], dtype=torch.float64)\n"
        "dist.all_reduce(t)\n"
        "open(sys.argv[1] + '.' + os.environ['RANK'], 'w').write(json.dumps(dict(rank=int(os.environ['RANK']), local=int(os.environ['LOCAL_RANK']),\n"
        "    world=int(os.environ['WORLD_SIZE']), addr=os.environ['MASTER_ADDR'], total=float(t.item()))))\n"
        "dist.destroy_process_group()\n"
        "sys.exit(int(sys.argv[2]) if os.environ['RANK'] == '1' else 0)\n")
    out = str(tmp_path / "res")
    assert bench.launch_ranks(3, [out, "0"], script=str(child)) == 0
    res = [json.load(open("%s.%d" % (out, r))) for r in range(3)]
    assert [r["rank"] for r in res] == [0, 1, 2] and [r["local"] for r in res] == [0, 1, 2]
    assert all(r["world"] == 3 and r["addr"] == "127.0.0.1" and r["total"] == 3.0 for r in res)
    assert bench.launch_ranks(2, [out, "7"], script=str(child)) == 7      # a failing rank's exit code is returned


def test_function_model_follows_the_wrapper():
    """moi_evaluator.FunctionModel against src/MOI_wrapper.jl:683-1012 on a hand-worked model: block order of the rows, the
    Jacobian pattern with the duplicates quadratic terms emit, the 1/2 convention of diagonal quadratic terms, bounds, sense."""
    from activesetmethods_amd.moi_evaluator import FunctionModel, ScalarFunction
    fm = FunctionModel(3, [-1, -1, -1], [2, 2, 2])
    fm.add_constraint(ScalarFunction(1.0, [], [(2.0, 1, 1), (3.0, 1, 2)]), "le", 5.0)           # quadratic <= : 1 + x1^2 + 3 x1 x2
    fm.add_constraint(ScalarFunction(0.0, [(1.0, 2), (2.0, 3)]), "eq", 1.0)                      # linear ==
    fm.add_constraint(ScalarFunction(0.5, [(4.0, 1)]), "ge", 0.0)                                # linear >=
    fm.objective = ScalarFunction(0.0, [(1.0, 3)], [(4.0, 2, 2), (1.0, 1, 3)])                   # x3 + 2 x2^2 + x1 x3
    fm.sense = "MAX_SENSE"
    # rows: linear >= (0.5 + 4 x1), linear == (x2 + 2 x3), quadratic <=          MOI_wrapper.jl:683-689
    assert fm.jacobian_structure() == [(1, 1), (2, 2), (2, 3), (3, 1), (3, 1), (3, 2)]          # :693-746 (x1^2 -> one entry, x1 x2 -> two)
    lb, ub = fm.constraint_bounds()
    assert list(lb) == [0.0, 1.0, -np.inf] and list(ub) == [np.inf, 1.0, 5.0]
    x = np.array([1.0, 2.0, 3.0])
    g = fm.eval_g(x, np.zeros(3))
    assert list(g) == [4.5, 8.0, 1.0 + 1.0 + 6.0]
    assert list(fm.eval_jac_g(x, np.zeros(6))) == [4.0, 1.0, 2.0, 2.0 * 1.0, 3.0 * 2.0, 3.0 * 1.0]   # :889-918
    assert fm.eval_f(x) == -(3.0 + 8.0 + 3.0)                                                    # MAX: scale -1 (:1037-1049)
    assert list(fm.eval_grad_f(x, np.zeros(3))) == [-3.0, -8.0, -2.0]
    assert list(fm.start_point()) == [0.0, 0.0, 0.0]
    pr = fm.to_problem()
    assert pr.m == 3 and pr.nnz == 6 and list(pr.j_row) == [1, 2, 2, 3, 3, 3]


@pytest.mark.parametrize("seed,sense", [(1, "MIN_SENSE"), (2, "MAX_SENSE"), (3, "FEASIBILITY_SENSE")])
def test_host_evaluator_equals_the_oracle_restatement(seed, sense):
    """The product's host evaluator (activesetmethods_amd/moi_evaluator.py) against oracle/moi_eval.py - two restatements of
    MOI_wrapper.jl:683-944 that share no code: pattern, f, grad f, g and the Jacobian values bit for bit."""
    from tests.util import random_function_model, oracle_wrapper_model, oracle_evaluate
    fm = random_function_model(seed, sense=sense)
    pr = fm.to_problem()
    om = oracle_wrapper_model(fm)
    rng = np.random.default_rng(seed + 7)
    for _ in range(3):
        x = rng.uniform(-1.0, 1.0, pr.n)
        fo, go, Eo, dEo, j_str = oracle_evaluate(om, x)
        assert [tuple(t) for t in zip(pr.j_row, pr.j_col)] == j_str
        assert pr.eval_f(x) == fo and np.array_equal(pr.eval_grad_f(x, np.zeros(pr.n)), go)
        assert np.array_equal(pr.eval_g(x, np.zeros(pr.m)), Eo) and np.array_equal(pr.eval_jac_g(x, np.zeros(pr.nnz)), dEo)


def _claim_worker(rank, world, port, q):
    import time as _t
    import torch.distributed as dist
    from activesetmethods_amd.batch import claim_chunks
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    got = []
    for lo, hi in claim_chunks(23, 4):
        got.append((lo, hi))
        _t.sleep(0.002 * (1 + 3 * rank))            # the ranks work at different speeds: the faster one claims more
    dist.barrier()
    q.put((rank, got))
    dist.destroy_process_group()


def test_dynamic_scenario_assignment_gloo_world2():
    """`claim_chunks`: the ranks of a batch claim chunks of scenario indices from one counter in the process group's store (world 2, gloo):
    every index is claimed exactly once, the last chunk is cut at the total, and the faster rank gets more."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_claim_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    spans = sorted(res[0] + res[1])
    assert spans[0][0] == 0 and spans[-1][1] == 23 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert all(hi - lo == 4 for lo, hi in spans[:-1]) and spans[-1][1] - spans[-1][0] == 3
    assert len(res[0]) >= len(res[1]) and len(res[0]) + len(res[1]) == 6
    # without a process group: one local counter
    from activesetmethods_amd.batch import claim_chunks
    assert list(claim_chunks(10, 4)) == [(0, 4), (4, 8), (8, 10)]
