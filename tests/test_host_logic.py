"""CPU tests of the host-side logic that needs no GPU: group normalisation, indicator vectors, mappings,
the SPG driver (on numpy), synthetic-input bookkeeping."""
import numpy as np
import pytest

from bluest_amd import synth


def test_algorithmic_bytes_match_survey():
    assert synth.algorithmic_bytes(12, 12)["eval"] == 3014736
    assert synth.algorithmic_bytes(20, 5)["eval"] == 9577424
    assert synth.algorithmic_bytes(25, 6)["eval"] == 152961080
    assert synth.n_groups(20, 5) == 21699 and synth.n_groups(25, 6) == 245505 and synth.n_groups(12, 12) == 4095


def test_indicator_vectors_and_mappings():
    from bluest_amd.mosap import build_mappings
    from bluest_amd.sap import indicator_vectors, normalise_groups
    groups = [[(0,), (1,), (2,)], [(0, 1), (0, 2), (1, 2)], [(0, 1, 2)]]
    flat = normalise_groups(groups, 3)
    assert flat == [[0], [1], [2], [0, 1], [0, 2], [1, 2], [0, 1, 2]]
    assert all(isinstance(g, np.ndarray) and g.dtype == np.int64 for g in groups)
    ES = indicator_vectors(groups, 3)
    assert (ES[0] == [1, 0, 0, 1, 1, 0, 1]).all() and (ES[2] == [0, 0, 1, 0, 1, 1, 1]).all()
    from bluest_amd.sap import LazyIndicators
    lazy = LazyIndicators(groups, 3)
    assert lazy._all is None and (lazy[0] == ES[0]).all() and lazy._all is None      # row 0 without building the table
    assert (lazy[2] == ES[2]).all() and len(lazy) == 3 and all((a == b).all() for a, b in zip(lazy, ES))
    cum = np.cumsum([0, 3, 3, 1])
    mg = [[np.array([[0], [2]]), np.array([[1, 2]]), np.zeros((0, 3), dtype=np.int64)],
          [np.array([[1]]), np.array([[0, 1], [0, 2]]), np.array([[0, 1, 2]])]]
    maps = build_mappings(groups, mg, cum, 3)
    assert maps[0].tolist() == [0, 2, 5] and maps[1].tolist() == [1, 3, 4, 6]
    with pytest.raises(AssertionError):
        build_mappings(groups, [[np.array([[2], [1], [0]])[:0], np.array([[1, 0]])]], cum, 3)    # (1,0) is not a listed group
    # large groups fall back to tuple keys (N^k would overflow int64)
    big = [np.zeros((0, k), dtype=np.int64) for k in range(1, 16)] + [np.array([np.arange(16), np.arange(1, 17)])]
    cumb = np.cumsum([0] + [len(g) for g in big])
    assert build_mappings(big, [[np.zeros((0, k), dtype=np.int64) for k in range(1, 16)] + [np.array([np.arange(1, 17)])]], cumb, 64)[0].tolist() == [1]


def test_spg_driver_on_numpy(oracle):
    """same driver as on the GPU, numpy vectors: box-simplex QP with known solution; identical to the oracle's spg"""
    from bluest_amd.spg import spg
    rng = np.random.RandomState(0)
    A = rng.randn(30, 30); A = A @ A.T + np.eye(30)
    b = rng.randn(30)
    feval = lambda x: 0.5 * x @ A @ x - b @ x
    geval = lambda x: A @ x - b
    proj = oracle.simplex_projection
    x0 = np.ones(30) / 30
    r1 = spg(feval, geval, proj, x0, eps=1e-10, maxit=500, verbose=False)
    r2 = oracle.spg(feval, geval, proj, x0, eps=1e-10, maxit=500)
    assert r1["solver_info"] == 0 and r1["it"] == r2["it"] and r1["count"] == r2["count"]
    assert np.array_equal(r1["x"], r2["x"]) and r1["f"] == r2["f"]
    # KKT: gradient constant on the support, larger off it
    g = geval(r1["x"]); sup = r1["x"] > 0
    assert np.ptp(g[sup]) < 1e-8 and (g[~sup] >= g[sup].max() - 1e-8).all()


def test_host_section_limits_blas_and_pauses_the_collector():
    """bluest_amd/host.py: inside the section the automatic cyclic collector is off and every BLAS pool runs on one thread (a
    container whose CPU quota is below its visible core count gets throttled by OpenBLAS's spinning workers otherwise,
    profiles/r02_host_stall.txt); both are restored on exit, also when the body raises, and sections nest"""
    import gc
    from bluest_amd.host import host_section, in_host_section
    threadpoolctl = pytest.importorskip("threadpoolctl")

    def blas_threads():
        return [p["num_threads"] for p in threadpoolctl.threadpool_info() if p["user_api"] == "blas"]

    before = blas_threads()
    assert gc.isenabled() and before
    with host_section():
        assert not gc.isenabled() and all(n == 1 for n in blas_threads())
        with host_section():
            assert not gc.isenabled() and all(n == 1 for n in blas_threads())
        assert not gc.isenabled() and all(n == 1 for n in blas_threads())
    assert gc.isenabled() and blas_threads() == before

    @in_host_section
    def body(x):
        assert not gc.isenabled()
        raise KeyError(x)

    with pytest.raises(KeyError):
        body(3)
    assert gc.isenabled() and blas_threads() == before
    assert body.__name__ == "body"


def test_first_host_section_touches_no_file_and_imports_nothing():
    """round-2 driver run: `sap_wallclock.cold.setup_s` = 1.0 s on a freshly leased box whose image was still paging in, against
    12 ms on a warm one -- the first host_section imported threadpoolctl and scanned the loaded libraries inside the timed
    constructor.  After `import bluest_amd.sap` a fresh subprocess must enter its first section, and run the host half of
    MOSAP's set-up, without opening a file or importing a module."""
    import subprocess
    import sys
    code = r'''
import sys
import numpy as np
import bluest_amd.sap, bluest_amd.mosap
from bluest_amd import synth
from bluest_amd.host import host_section
from bluest_amd.mosap import build_mappings
from bluest_amd.sap import LazyIndicators, normalise_groups
prob = synth.problem(8, 3, 2)
events = []
sys.addaudithook(lambda ev, args: events.append((ev, str(args)[:120])) if ev in ("open", "import", "ctypes.dlopen") else None)
with host_section():
    groups = [g.copy() for g in prob["groups"]]
    normalise_groups(groups, 3, flatten=False)
    e = LazyIndicators(groups, 8)[0]
    maps = build_mappings(groups, [[g.copy() for g in groups]] * 2, np.cumsum([0] + [len(g) for g in groups]), 8)
    tot = (np.ones(len(e)) @ e)
bad = [ev for ev in events if ev[0] != "import" or "None" in ev[1]]      # "import" of an already loaded module still reports
bad = [ev for ev in events if ev[0] in ("open", "ctypes.dlopen")]
print("EVENTS", bad)
assert not bad, bad
'''
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=__import__("os").path.dirname(__import__("os").path.dirname(__file__)), timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr


def test_blueproblem_mpi_semantics_without_mpi():
    """with a communicator of several ranks only rank 0 owns a MOSAP; the other ranks must get the estimators through
    bcast (bluest/blue_models.py:565-571), the samples are split over the ranks and the sums all-reduced (blue_fn.py:107-111,
    :178-182); without a communicator get_comm() is a one-rank stand-in"""
    from bluest_amd.blue_models import BLUEProblem

    class FakeComm(object):
        def __init__(self, rank, size, root_values=None):
            self.rank, self.size, self.root_values, self.reduced = rank, size, list(root_values or []), []
        def Get_rank(self): return self.rank
        def Get_size(self): return self.size
        def bcast(self, obj, root=0):
            return obj if self.rank == root else self.root_values.pop(0)
        def allreduce(self, obj, op=None):
            self.reduced.append(obj)
            return obj * self.size          # pretend every rank contributed the same

    class P(BLUEProblem):
        calls = 0
        def sampler(self, ls, N=1): return [0.5 for _ in ls]
        def evaluate(self, ls, samples):
            P.calls += 1
            return [[1.0 for _ in ls] for _ in range(self.n_outputs)]

    C = np.eye(3) + 0.5
    serial = P(3, C=C, costs=np.array([4.0, 2.0, 1.0]), verbose=False)
    assert serial.get_comm().Get_rank() == 0 and serial.get_comm().Get_size() == 1 and serial.get_comm().bcast(7) == 7
    assert serial._group_sums([0, 2], 5) == [[5.0, 5.0]]
    # 3 ranks, 7 samples: 3 + 2 + 2
    counts = []
    for r in range(3):
        P.calls = 0
        p = P(3, C=C, costs=np.array([4.0, 2.0, 1.0]), verbose=False, comm=FakeComm(r, 3))
        sums = p._group_sums([0, 1], 7)
        counts.append(P.calls)
        assert sums == [[3.0 * P.calls, 3.0 * P.calls]] and len(p.comm.reduced) == 2
    assert counts == [3, 2, 2]
    # a non-root rank has no MOSAP: solve() takes the estimators from the broadcast
    comm = FakeComm(1, 2, root_values=[[1.25], np.array([4.0])])
    p = P(3, C=C, costs=np.array([4.0, 2.0, 1.0]), verbose=False, comm=comm)
    p.MOSAP_output = {"budget": 10.0, "eps": None, "samples": np.array([2.0, 0.0]), "flattened_groups": [[0], [1]],
                      "variances": [4.0], "cost": 8.0}
    mus, errs, cost = p.solve(budget=10.0)
    assert p.MOSAP is None and mus == [1.25] and errs[0] == 2.0 and cost == 8.0


def test_sample_caps_by_shifted_costs_on_a_separable_problem(monkeypatch):
    """the outer iteration of bluest_amd.capped.cost_shift_capped (shifted costs, brackets, Dantzig-Wolfe prices, primal
    recovery, certified gap) with the free solver replaced by a closed form: F(m) = sum_i v_i / m_i is convex and homogeneous of
    degree -1 like the estimator variance, its free optimum under costs w and budget B is m_i = B sqrt(v_i / w_i) / sum_j sqrt(v_j w_j),
    and with caps m_i <= n_i on single entries the optimum is: capped entries at their caps, the rest of the budget spent freely"""
    import torch
    from bluest_amd import capped, colgen
    rng = np.random.RandomState(3)
    L = 12
    v, w, B = 0.5 + rng.rand(L), 0.2 + rng.rand(L), 100.0

    def free(costs, budget):
        m = budget * np.sqrt(v / costs) / np.sqrt(v * costs).sum()
        return m, float((v / m).sum())

    def fake_colgen(plan, costs, s, budget, x0=None, prm=None, log=None, caps=None):
        m, F = free(np.asarray(costs), budget)
        x = np.asarray(costs) * m / budget
        return x, {"F": F, "lower_bound": F * (1 - 1e-12), "gap": 1e-12, "newton_it": 1, "full_evals": 1, "master_evals": 0, "rounds": 1, "mu": np.ones(1)}

    class FakePlan(object):
        n_out = 1

        def eval(self, m, want_grad=False):
            return torch.tensor([[float((v / np.asarray(m)).sum())]], dtype=torch.float64), None, torch.zeros((1, 1), dtype=torch.int32)

    monkeypatch.setattr(colgen, "colgen_solve", fake_colgen)
    m_free, F_free = free(w, B)
    capped_idx = np.argsort(-m_free)[:3]
    rows = np.zeros((3, L))
    rows[np.arange(3), capped_idx] = 1.0
    rhs = np.array([0.5, 0.6, 0.7]) * m_free[capped_idx]
    m, info = capped.cost_shift_capped(FakePlan(), w, np.ones(1), B, rows, rhs)
    assert m is not None, info
    # closed form: the capped entries sit at their caps, the others share what is left of the budget
    rest = np.setdiff1d(np.arange(L), capped_idx)
    B_rest = B - float(w[capped_idx] @ rhs)
    m_ref = np.zeros(L)
    m_ref[capped_idx] = rhs
    m_ref[rest] = B_rest * np.sqrt(v[rest] / w[rest]) / np.sqrt(v[rest] * w[rest]).sum()
    F_ref = float((v / m_ref).sum())
    assert m @ w <= B * (1 + 1e-12) and (rows @ m <= rhs * (1 + 1e-12)).all()
    assert abs(info["F"] / float((v / m).sum()) - 1) < 1e-12
    assert info["lower_bound"] <= F_ref * (1 + 1e-9) and info["F"] >= F_ref * (1 - 1e-9)      # bound and value bracket the optimum
    assert info["F"] / F_ref - 1 < 1e-5 and info["gap"] < 1e-5, (info["F"], F_ref, info["gap"], info["solves"])
    assert np.abs(m - m_ref).max() < 1e-2 * m_ref.max()
    # caps that do not bind: the free optimum comes back after one solve
    m2, info2 = capped.cost_shift_capped(FakePlan(), w, np.ones(1), B, rows, 2.0 * m_free[capped_idx])
    assert info2["solves"] == 1 and np.allclose(m2, m_free, rtol=1e-12)


def test_estimator_rhs_matches_the_group_by_group_accumulation():
    """SAP.compute_BLUE_estimator's right-hand side y = sum_i R_i^T C_i^-1 sums_i (bluest/sap.py:104-110): the vectorised
    accumulation (einsum + np.add.at per group size) against the plain quadruple loop, for scalar sums, array-valued sums and
    sums that are opaque objects; unsampled groups contribute nothing."""
    import itertools
    from fractions import Fraction
    from bluest_amd.sap import estimator_rhs
    rng = np.random.RandomState(5)
    N, K = 5, 3
    groups = [np.array(list(itertools.combinations(range(N), k)), dtype=np.int64) for k in range(1, K + 1)]
    sizes = [0] + [len(g) for g in groups]
    cs = np.cumsum(sizes)
    invcovs = [rng.randn(len(groups[k - 1]) * k * k) for k in range(1, K + 1)]
    L = int(cs[-1])
    m = np.where(rng.rand(L) < 0.6, rng.randint(1, 9, L), 0).astype(float)

    def loop(sums):
        y = [0.0] * N
        for k in range(1, K + 1):
            for i in range(sizes[k]):
                if m[cs[k - 1] + i] == 0:
                    continue
                for j in range(k):
                    for s in range(k):
                        y[groups[k - 1][i][j]] = y[groups[k - 1][i][j]] + invcovs[k - 1][k * k * i + k * j + s] * sums[cs[k - 1] + i][s]
        return y

    flat_k = [k for k in range(1, K + 1) for _ in range(sizes[k])]
    scalar = [list(rng.randn(k)) for k in flat_k]
    y_num, y_obj = estimator_rhs(N, K, cs, groups, invcovs, scalar, m)
    assert not y_obj and np.allclose(y_num, loop(scalar), rtol=1e-13, atol=1e-13)
    vec = [[rng.randn(3) for _ in range(k)] for k in flat_k]
    y_num, y_obj = estimator_rhs(N, K, cs, groups, invcovs, vec, m)
    assert not y_obj and y_num.shape == (N, 3) and np.allclose(y_num, np.array(loop(vec)), rtol=1e-13, atol=1e-13)

    class Opaque(object):                                   # a user object that only knows + and scalar *
        def __init__(self, v): self.v = v
        def __add__(self, o): return Opaque(self.v + o.v)
        def __rmul__(self, c): return Opaque(c * self.v)
    obj = [[Opaque(v) for v in row] for row in scalar]
    y_num, y_obj = estimator_rhs(N, K, cs, groups, invcovs, obj, m)
    assert y_num is None
    ref = loop(scalar)
    for j in range(N):
        assert abs(y_obj[j].v - ref[j]) <= 1e-12 * max(1.0, abs(ref[j]))
    assert Fraction(1, 2) + 0 == Fraction(1, 2)


def test_blue_fn_sums_and_refusals():
    """bluest_amd.blue_fn (role of bluest/blue_fn.py:36-211): sums of the outputs and of their pairwise inner products, the batched
    sampler gives the same sums as one path at a time, a stated problem cost wins, the sample-file options are refused"""
    import bluest_amd

    class Problem(object):
        def evaluate(self, ls, inputs):
            return [[inputs[i] * (l + 1) for i, l in enumerate(ls)], [inputs[i] ** 2 + l for i, l in enumerate(ls)]]

    ls = [0, 2]
    se, sc, cost = bluest_amd.blue_fn(ls, 40, Problem(), No=2, verbose=False)
    draws = np.random.RandomState(1).randn(40)
    assert np.allclose(np.ravel(se[0]), [draws.sum(), 3 * draws.sum()])
    assert np.allclose(sc[0], np.array([[1, 3], [3, 9]]) * (draws ** 2).sum())
    assert np.allclose(np.ravel(se[1]), [(draws ** 2).sum(), (draws ** 2 + 2).sum()])
    assert cost >= 0.0
    se7, sc7, _ = bluest_amd.blue_fn(ls, 40, Problem(), No=2, N1=7, verbose=False)
    assert np.allclose(np.ravel(se7[0]), np.ravel(se[0])) and np.allclose(sc7[1], sc[1])
    Problem.cost = 2.5
    assert bluest_amd.blue_fn(ls, 4, Problem(), No=2, verbose=False)[2] == 10.0
    with pytest.raises(bluest_amd.BLUESTError):
        bluest_amd.blue_fn(ls, 4, Problem(), No=2, filename="x.npz")


def test_head_of_the_candidate_list_equals_the_full_sort():
    """colgen.top_candidates (the pricing round's host side: a partition + a sort of the head instead of np.lexsort over all 1024
    candidates) returns exactly the head of the full (largest value first, ties by smaller index) order -- with ties across the cut,
    empty slots (index -1), and k beyond the list"""
    from bluest_amd.colgen import top_candidates
    rng = np.random.RandomState(7)
    for trial in range(300):
        n = int(rng.choice([16, 64, 1024]))
        topv = np.round(rng.rand(n), int(rng.choice([1, 2, 6])))          # coarse rounding: many ties
        topi = rng.randint(-1, 4 * n, n).astype(np.int64)
        if trial % 7 == 0:
            topv[rng.rand(n) < 0.5] = -np.inf
        k = int(rng.randint(1, n + 40))
        order = np.lexsort((topi, -topv))
        ci, cv = top_candidates(topv, topi, k)
        m = min(k, n)
        assert ci[:m] == topi[order][:m].tolist() and cv[:m] == topv[order][:m].tolist(), (trial, n, k)
