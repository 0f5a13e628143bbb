"""
GPU tests of the second-order finish (C-ABI Part 6, csrc/newton.hip) against its numpy restatement (oracle/master_newton.py)
and against oracle-evaluated certificates.
"""
import ctypes

import numpy as np
import pytest

from bluest_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("these tests need the GPU (run with -m gpu on the MI355X box)")
    return torch


def _support_problem(oracle, prob, keep, eps_bg, s):
    """SupportProblem (numpy restatement) of a synthetic problem restricted to the groups `keep`"""
    from oracle.master_newton import SupportProblem
    n, kmax, n_out = prob["n"], prob["kmax"], prob["n_out"]
    saps = [oracle.SparseOracleSAP(C, kmax, prob["groups"]) for C in prob["C"]]
    sp0 = saps[0]
    flat = [g for gk in sp0.groups for g in gk]
    c = prob["budget"] / prob["costs"]

    def block(o, i):
        k = int(np.searchsorted(sp0.cumsizes, i, side="right"))
        return saps[o].invcovs[k - 1].reshape(-1, k, k)[i - sp0.cumsizes[k - 1]]
    phi_u = np.array([q.get_phi(c / sp0.L) for q in saps])
    sp = SupportProblem(n, [flat[i] for i in keep], [[block(o, i) for i in keep] for o in range(n_out)], c[keep], s, eps_bg * phi_u, eps_bg)
    return sp, phi_u, saps


def _gpu_master(torch, plan, keep, cc_keep, s, bg, eps_bg, x0, mu0, tol=1e-10, maxit=60):
    from bluest_amd._lib import check
    from bluest_amd.plan import _stream
    dev = plan.device
    n_out = plan.n_out
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)      # noqa: E731
    x_d, mu_d, s_d = to_dev(x0), to_dev(mu0), to_dev(s)
    out_d = torch.zeros(16 + n_out, dtype=torch.float64, device=dev)
    bg_d = None if bg is None else to_dev(bg)
    keep = np.ascontiguousarray(keep, dtype=np.int64)
    cc_keep = np.ascontiguousarray(cc_keep, dtype=np.float64)
    with torch.cuda.device(dev):
        check(plan.lib.bluest_master_newton(plan._h, len(keep), keep.ctypes.data, cc_keep.ctypes.data, s_d.data_ptr(),
                                            None if bg_d is None else bg_d.data_ptr(), float(eps_bg), x_d.data_ptr(), mu_d.data_ptr(),
                                            float(tol), int(maxit), out_d.data_ptr(), _stream()))
    torch.cuda.synchronize()
    return x_d.cpu().numpy(), mu_d.cpu().numpy(), out_d.cpu().numpy()


@pytest.mark.parametrize("n,kmax,n_out,S,eps_bg", [(8, 3, 1, 20, 1e-3), (8, 3, 2, 24, 1e-3), (10, 4, 3, 30, 1e-6), (20, 5, 8, 48, 1e-3)])
def test_master_kernel_equals_numpy_restatement(gpu, oracle, n, kmax, n_out, S, eps_bg):
    """one master problem from the same start: the single-workgroup kernel and the numpy restatement end at the same optimum
    (objective to 1e-9, allocation to 1e-5, multipliers to 1e-4), both with a KKT residual below 1e-6; the headline shape runs at
    the largest support its LDS budget takes (bluest_master_max_support)"""
    from oracle.master_newton import master_newton
    from bluest_amd.plan import Plan
    prob = synth.problem(n, kmax, n_out)
    sizes = [len(g) for g in prob["groups"]]
    plan = Plan(n, prob["K_tot"], [{"K": kmax, "sizes": sizes, "groups": prob["groups"], "C": prob["C"][o], "mapping": None} for o in range(n_out)])
    rng = np.random.RandomState(5)
    from bluest_amd.colgen import master_max_support
    assert master_max_support(plan) >= S
    keep = np.sort(np.concatenate([[0], 1 + rng.choice(prob["K_tot"] - 1, S - 1, replace=False)]))
    if eps_bg == 0.0:                                   # without background every model must be sampled by the support itself
        keep = np.union1d(keep, np.arange(n))
    s = 1.0 + 0.3 * rng.rand(n_out) if n_out > 1 else np.ones(1)
    sp, phi_u, saps = _support_problem(oracle, prob, keep, eps_bg, s)
    x0 = rng.rand(len(keep)) + 0.1
    x0 /= x0.sum()
    mu0 = np.full(n_out, 1.0 / n_out)
    ref = master_newton(sp, x0, mu0=mu0, tol=1e-10)
    assert ref["kkt"] <= 1e-6            # both stop at the tolerance or where the objective cannot resolve further progress
    cc = prob["budget"] / prob["costs"]
    x, mu, out = _gpu_master(gpu, plan, keep, cc[keep], s, None if eps_bg == 0.0 else eps_bg * phi_u, eps_bg, x0, mu0)
    assert int(out[7]) == 0 and out[2] <= 1e-6, out[:10]
    assert abs(out[0] / ref["F"] - 1) < 1e-9
    assert np.abs(out[16:] / ref["r"] - 1).max() < 1e-8
    assert np.abs(x - ref["x"]).max() < 1e-5 and abs(x.sum() - 1) < 1e-12 and x.min() >= 0
    assert np.abs(mu - ref["mu"]).max() < 1e-4
    # the kernel's objective is what the plan itself evaluates at that allocation (background included)
    m = np.zeros(prob["K_tot"])
    m[keep] = cc[keep] * x
    m = (1 - eps_bg) * m + eps_bg * cc / prob["K_tot"]
    var, _, st = plan.eval(m, want_grad=False)
    assert (st.cpu().numpy() == 0).all()
    assert abs((var[0].cpu().numpy() / s).max() / out[0] - 1) < 1e-10


@pytest.mark.parametrize("n,kmax,n_out,S", [(8, 3, 1, 20), (10, 4, 2, 30), (20, 5, 8, 48)])
def test_capped_master_kernel_equals_numpy_restatement(gpu, oracle, n, kmax, n_out, S):
    """the master with per-model sample caps (bluest_master_newton_capped; max_model_samples, bluest/sap.py:222-240): same optimum as
    the numpy restatement (itself checked against SLSQP in tests/test_oracle.py), caps respected, multipliers >= 0"""
    from oracle.master_newton import master_newton
    from bluest_amd._lib import check
    from bluest_amd.plan import Plan, _stream
    torch = gpu
    prob = synth.problem(n, kmax, n_out)
    sizes = [len(g) for g in prob["groups"]]
    plan = Plan(n, prob["K_tot"], [{"K": kmax, "sizes": sizes, "groups": prob["groups"], "C": prob["C"][o], "mapping": None} for o in range(n_out)])
    rng = np.random.RandomState(5)
    keep = np.sort(np.concatenate([[0], 1 + rng.choice(prob["K_tot"] - 1, S - 1, replace=False)]))
    s = 1.0 + 0.3 * rng.rand(n_out) if n_out > 1 else np.ones(1)
    eps_bg = 1e-3
    sp, phi_u, saps = _support_problem(oracle, prob, keep, eps_bg, s)
    flat = [g for gk in prob["groups"] for g in gk]
    cc = prob["budget"] / prob["costs"]
    x0 = np.full(S, 1.0 / S)
    ref0 = master_newton(sp, x0, tol=1e-10)
    usage = np.array([sum(cc[i] * ref0["x"][j] for j, i in enumerate(keep) if mdl in flat[i]) for mdl in range(n)])
    cap_models = [0, 1 + int(np.argmax(usage[1:]))]          # model 0 and the most sampled other model: both caps bind
    Acap = np.array([[cc[i] * (1 - eps_bg) if mdl in flat[i] else 0.0 for i in keep] for mdl in cap_models])
    bcap = 0.6 * (Acap @ ref0["x"])
    x = x0.copy()
    for _ in range(200):
        viol = Acap @ x - bcap
        if (viol <= 0).all():
            break
        c_ = int(np.argmax(viol / bcap))
        msk = Acap[c_] > 0
        x[msk] *= 0.95 * bcap[c_] / (Acap[c_] @ x)
        x[~msk] += (1 - x.sum()) * x[~msk] / x[~msk].sum()
    ref = master_newton(sp, x, tol=1e-10, caps=(Acap, bcap))
    dev = plan.device
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)      # noqa: E731
    x_d, mu_d, s_d, bg_d = to_dev(x), to_dev(np.full(n_out, 1.0 / n_out)), to_dev(s), to_dev(eps_bg * phi_u)
    nu_d = torch.zeros(2, dtype=torch.float64, device=dev)
    out_d = torch.zeros(16 + n_out, dtype=torch.float64, device=dev)
    keep64, cck = np.ascontiguousarray(keep, dtype=np.int64), np.ascontiguousarray(cc[keep])
    cm, cb = np.asarray(cap_models, dtype=np.int32), np.ascontiguousarray(bcap)
    with torch.cuda.device(dev):
        check(plan.lib.bluest_master_newton_capped(plan._h, S, keep64.ctypes.data, cck.ctypes.data, s_d.data_ptr(), bg_d.data_ptr(), eps_bg,
                                                   x_d.data_ptr(), mu_d.data_ptr(), 1e-10, 60, out_d.data_ptr(), 2, cm.ctypes.data, cb.ctypes.data,
                                                   nu_d.data_ptr(), _stream()))
    torch.cuda.synchronize()
    xg, out, nu = x_d.cpu().numpy(), out_d.cpu().numpy(), nu_d.cpu().numpy()
    # status 1 (no step accepted any more) is as good an end as running out of iterations where neither version converges (below)
    assert int(out[7]) in (0, 1) and np.isfinite(out[0]), out[:10]
    assert (Acap @ xg <= bcap * (1 + 1e-9)).all() and nu.min() >= 0 and abs(xg.sum() - 1) < 1e-12 and xg.min() >= 0
    assert out[0] > ref0["F"] * (1 + 1e-6)                      # the caps bind
    print("capped master: kernel F %.12e kkt %.2e it %d evals %d | numpy F %.12e kkt %.2e it %d" % (out[0], out[2], out[4], out[5], ref["F"], ref["kkt"], ref["it"]))
    # (with several outputs AND caps both versions can use up the 60 iterations at a KKT residual of 1e-2: the same point then
    # only to ~1e-4; the column generation keeps iterating from there)
    tol = 1e-7 if (out[2] <= 1e-6 and ref["kkt"] <= 1e-6) else 2e-4
    assert abs(out[0] / ref["F"] - 1) < tol, (out[0], ref["F"], out[:10], ref["kkt"])


def test_master_kernel_with_ragged_outputs(gpu, oracle):
    """outputs with their own group subsets (non-identity mappings, bluest/mosap.py:54-67): a support group that an output does not
    use has no block for it (boff = -1 in the kernel, None in the restatement); same optimum"""
    from oracle.master_newton import SupportProblem, master_newton
    from bluest_amd.plan import Plan
    n, kmax, n_out, S, eps_bg = 9, 3, 3, 28, 1e-3
    prob = synth.problem(n, kmax, n_out)
    groups = prob["groups"]
    L = prob["K_tot"]
    sizes = [len(g) for g in groups]
    rng = np.random.RandomState(11)
    cum = np.cumsum([0] + sizes)
    outs, keeps = [], []
    for o in range(n_out):
        keep_o = [np.ones(len(g), dtype=bool) if o == 0 else rng.rand(len(g)) < 0.6 for g in groups]
        for kp in keep_o:
            kp[0] = True
        keeps.append(np.concatenate(keep_o))
        outs.append({"K": kmax, "sizes": [int(kp.sum()) for kp in keep_o], "groups": [g[kp] for g, kp in zip(groups, keep_o)], "C": prob["C"][o],
                     "mapping": None if o == 0 else np.flatnonzero(keeps[-1])})
    plan = Plan(n, L, outs)
    keep = np.sort(np.concatenate([[0], 1 + rng.choice(L - 1, S - 1, replace=False)]))
    s = np.ones(n_out)
    saps = [oracle.SparseOracleSAP(C, kmax, groups) for C in prob["C"]]            # blocks of every group; masked below
    sp0 = saps[0]
    flat = [g for gk in sp0.groups for g in gk]
    cc = prob["budget"] / prob["costs"]

    def block(o, i):
        if not keeps[o][i]:
            return None
        k = int(np.searchsorted(sp0.cumsizes, i, side="right"))
        return saps[o].invcovs[k - 1].reshape(-1, k, k)[i - sp0.cumsizes[k - 1]]
    u = cc / L
    phi_u = plan.phi(u).cpu().numpy()[0][:, :n * n].reshape(n_out, n, n)           # each output's own groups only
    for o in range(n_out):                                                       # ... which the oracle reproduces with masked m
        assert np.abs(phi_u[o] - saps[o].get_phi(np.where(keeps[o], u, 0.0))).max() < 1e-12 * np.abs(phi_u[o]).max()
    sp = SupportProblem(n, [flat[i] for i in keep], [[block(o, i) for i in keep] for o in range(n_out)], cc[keep], s, eps_bg * phi_u, eps_bg)
    x0 = rng.rand(S) + 0.1
    x0 /= x0.sum()
    ref = master_newton(sp, x0, tol=1e-10)
    x, mu, out = _gpu_master(gpu, plan, keep, cc[keep], s, eps_bg * phi_u, eps_bg, x0, np.full(n_out, 1.0 / n_out))
    print("ragged master: kernel F %.12e kkt %.1e it %d status %d | numpy F %.12e kkt %.1e it %d" % (out[0], out[2], out[4], out[7], ref["F"], ref["kkt"], ref["it"]))
    assert int(out[7]) == 0 and abs(out[0] / ref["F"] - 1) < 1e-9 and np.abs(x - ref["x"]).max() < 1e-5


@pytest.mark.parametrize("n,kmax,n_out", [(2, 2, 1), (3, 1, 1), (3, 3, 2), (4, 2, 3), (5, 5, 1), (7, 1, 2)])
def test_second_order_finish_on_tiny_problems(gpu, oracle, n, kmax, n_out):
    """edge sizes of the column generation: fewer groups than the initial support, singletons only (k_max = 1: the optimum is a
    closed form only for one output, but the certificate must close anyway), more outputs than models.  Budget and eps mode;
    the result is compared with scipy's SLSQP on the ORACLE objective where the problem is small enough for it."""
    from scipy.optimize import minimize
    from bluest_amd.mosap import MOSAP
    prob = synth.problem(n, kmax, n_out)
    groups, w, B = prob["groups"], prob["costs"], prob["budget"]
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)], w, [w] * n_out,
                verbose=False)
    m = mos.solve(budget=B, solver="spg", continuous_relaxation=True)
    info = dict(mos.solver_info)
    assert m is not None and info.get("method") == "newton" and info["certified_gap"] <= 1e-6, info
    assert (m >= 0).all() and abs(m @ w / B - 1) < 1e-9
    refs = [oracle.OracleSAP(C, kmax, groups, w) for C in prob["C"]]
    F = max(r.variance(m) for r in refs)
    assert abs(max(mos.variances(m)) / F - 1) < 1e-10
    scale = B / w
    L = len(w)

    def fun(z):                                            # epigraph form for SLSQP: variables (x, t)
        return z[-1]
    cons = [{"type": "eq", "fun": lambda z: z[:-1].sum() - 1.0}]
    for r in refs:
        cons.append({"type": "ineq", "fun": lambda z, r=r: z[-1] - r.variance(scale * np.maximum(z[:-1], 0.0) + 1e-300) / F})
    z0 = np.concatenate([m / scale * 0.9 + 0.1 / L, [1.1]])
    sol = minimize(fun, z0, method="SLSQP", bounds=[(1e-9, 1.0)] * L + [(0.0, 10.0)], constraints=cons, options={"ftol": 1e-14, "maxiter": 300})
    if sol.status == 0:
        assert sol.fun >= 1.0 - 1e-5, (sol.fun, info)       # SLSQP cannot do better than the certified optimum
    # eps mode: every tolerance met, the binding one exactly
    eps = np.array([np.sqrt(C[0, 0]) / 15.0 for C in prob["C"]])
    m2 = mos.solve(eps=eps, solver="spg", continuous_relaxation=True)
    ratios = np.array(mos.variances(m2)) / eps ** 2
    assert m2 is not None and ratios.max() <= 1 + 1e-9 and abs(ratios.max() - 1) < 1e-9 and mos.solver_info["certified_gap"] <= 1e-6


@pytest.mark.parametrize("n,kmax,n_out", [(48, 2, 1), (36, 3, 4), (40, 3, 2), (30, 3, 4)])
def test_second_order_finish_beyond_32_models(gpu, n, kmax, n_out):
    """the default solver must not fall off a cliff at the sizes where the master's registers end (more than 32 models: Phi is
    eliminated out of LDS) or where its LDS budget bites (many outputs at large N): the second-order finish runs and its certified
    gap closes to 1e-5 or better (round 3: the first-order fall-back ended 1e-4 .. 3e-3 above the optimum there)"""
    from bluest_amd.mosap import MOSAP
    prob = synth.problem(n, kmax, n_out)
    groups, w, B = prob["groups"], prob["costs"], prob["budget"]
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)], w, [w] * n_out,
                verbose=False)
    m = mos.solve(budget=B, solver="spg", continuous_relaxation=True)
    info = dict(mos.solver_info)
    assert m is not None and info.get("method") == "newton", info
    assert info["certified_gap"] <= 1e-5, info
    assert (m >= 0).all() and abs(m @ w / B - 1) < 1e-9
    assert abs(max(mos.variances(m)) / info["f"] - 1) < 1e-9


@pytest.mark.parametrize("n,kmax,matfree", [(8, 3, "0"), (12, 12, "0"), (20, 5, "0"), (20, 5, "1"), (25, 6, "1"), (30, 3, "0"), (20, 5, "2"), (40, 2, "0")])
def test_multiplicative_update_inside_the_fused_kernel_is_bit_identical(gpu, monkeypatch, n, kmax, matfree):
    """single-output plans: bluest_plan_eval_ma (the tile wavefronts of the fused solve + gradient kernel apply the update) gives the
    iterates of bluest_plan_eval + bluest_ma_update bit for bit over 25 steps -- stored and matrix-free plans, s != 1 --, leaves the
    iterate alone when the allocation is not evaluable, and refuses plans with more than one output"""
    torch = gpu
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from bluest_amd._lib import BluestHipError, check
    from bluest_amd.plan import Plan, _stream
    monkeypatch.setenv("BLUEST_MATFREE", matfree)
    prob = synth.problem(n, kmax, 1)
    L = prob["K_tot"]
    plan = Plan(n, L, bench.build_outputs(prob))
    assert plan.matfree == (matfree == "1") and plan.matfree_gradient == (matfree in ("1", "2")) and plan.identity      # ("2": stored Phi pass, matrix-free gradient)
    dev = plan.device
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)      # noqa: E731
    cc_h = prob["budget"] / prob["costs"]
    cc, s_d = to_dev(cc_h), to_dev(np.array([0.37]))
    x0 = np.random.RandomState(5).rand(L) + 0.1
    x0 /= x0.sum()
    res = []
    with torch.cuda.device(dev):
        st = _stream()
        for fused in (False, True):
            x_d, m_d = to_dev(x0), to_dev(cc_h * x0)
            var = torch.empty((1, 1), dtype=torch.float64, device=dev)
            status = torch.empty((1, 1), dtype=torch.int32, device=dev)
            grad = torch.empty((1, plan.grad_len), dtype=torch.float64, device=dev)
            for _ in range(25):
                if fused:
                    check(plan.lib.bluest_plan_eval_ma(plan._h, m_d.data_ptr(), var.data_ptr(), status.data_ptr(), s_d.data_ptr(), cc.data_ptr(), x_d.data_ptr(), st))
                else:
                    check(plan.lib.bluest_plan_eval(plan._h, m_d.data_ptr(), 1, L, 0.0, var.data_ptr(), grad.data_ptr(), plan.grad_len, status.data_ptr(), st))
                    check(plan.lib.bluest_ma_update(plan._h, var.data_ptr(), status.data_ptr(), grad.data_ptr(), s_d.data_ptr(), cc.data_ptr(), 32.0,
                                                    x_d.data_ptr(), m_d.data_ptr(), st))
            torch.cuda.synchronize()
            res.append((x_d.cpu().numpy(), m_d.cpu().numpy(), var.cpu().numpy(), status.cpu().numpy()))
        (xa, ma, va, sa), (xb, mb, vb, sb) = res
        assert (sa == 0).all() and (sb == 0).all() and abs(xa.sum() - 1) < 1e-9 and (xa > 0).all()
        assert np.array_equal(va, vb) and np.array_equal(xa, xb) and np.array_equal(ma, mb)
        # not evaluable (model 0 unsampled): status says so, x and m stay as they were
        x1 = x0.copy()
        x1[np.asarray([0 in g for gk in prob["groups"] for g in gk])] = 0.0
        x_d, m_d = to_dev(x1), to_dev(cc_h * x1)
        check(plan.lib.bluest_plan_eval_ma(plan._h, m_d.data_ptr(), var.data_ptr(), status.data_ptr(), s_d.data_ptr(), cc.data_ptr(), x_d.data_ptr(), st))
        torch.cuda.synchronize()
        assert int(status.cpu()[0, 0]) != 0 and np.array_equal(x_d.cpu().numpy(), x1) and np.array_equal(m_d.cpu().numpy(), cc_h * x1)
        two = synth.problem(8, 3, 2)
        p2 = Plan(8, two["K_tot"], bench.build_outputs(two))
        with pytest.raises(BluestHipError):
            check(p2.lib.bluest_plan_eval_ma(p2._h, m_d.data_ptr(), var.data_ptr(), status.data_ptr(), s_d.data_ptr(), cc.data_ptr(), x_d.data_ptr(), st))
