"""
oracle/gen_golden.py -- produce tests/golden/*.npz by RUNNING THE REAL REFERENCE in the build container.

Runs only where /root/reference exists (never on the GPU box, never from tests).  It
  * builds oracle/_ref/_cmisc_bluest*.so from /root/reference/bluest/cmisc.cpp (oracle/Makefile `ref`),
  * imports the reference package from /root/reference with sys.dont_write_bytecode (nothing is copied);
    mpi4py / cvxpy / cvxopt are absent offline and are only touched by code OFF the hot path
    (blue_fn.py:9, sap.py:4,6, mosap.py:6,8), so empty placeholder modules are registered for the import
    (SURVEY.md section 8c),
  * evaluates the hot-path functions on seeded inputs and stores inputs + outputs as small fixtures.

A fixture is data only: inputs (or the seed that regenerates them through bluest_amd/synth.py) and the
reference's outputs.  The two paper data files copied verbatim (covariances / costs / stored allocation of the
Hodgkin-Huxley example) are data files the reference's own example drivers load.

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
import os
import subprocess
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from bluest_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (only for the build-defined simplex projection)


def import_reference():
    subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
    sys.path.insert(0, os.path.join(HERE, "_ref"))
    sys.path.insert(0, REF)

    class _Comm:
        def Get_size(self): return 1
        def Get_rank(self): return 0
        def bcast(self, x, root=0): return x
        def barrier(self): pass
        def allreduce(self, x, op=None): return x

    mpi = types.ModuleType("mpi4py"); MPI = types.ModuleType("mpi4py.MPI")
    MPI.COMM_WORLD = _Comm(); MPI.SUM = None; MPI.COMM_SELF = _Comm(); mpi.MPI = MPI
    cp = types.ModuleType("cvxpy"); cp.SolverError = type("SolverError", (Exception,), {})
    co = types.ModuleType("cvxopt"); co.matrix = co.spmatrix = co.solvers = None
    for name, mod in (("mpi4py", mpi), ("mpi4py.MPI", MPI), ("cvxpy", cp), ("cvxopt", co)):
        sys.modules.setdefault(name, mod)
    import _cmisc_bluest
    import bluest
    from bluest import misc, spg as spgmod
    return _cmisc_bluest, bluest, misc, spgmod


def lists_of(groups):
    """the reference constructors take list-of-lists and convert in place (sap.py:77)"""
    return [[list(map(int, g)) for g in gk] for gk in groups]


def gen_cmisc(cm):
    """direct known-answer vectors for every function of the native module (cmisc.cpp:99-110)"""
    rng = np.random.RandomState(7)
    N = 7
    out = {"N": N}
    for k, q in ((1, 2), (2, 3), (3, 3), (4, 2)):
        gk = np.array([sorted(rng.choice(N, k, replace=False)) for _ in range(23)], dtype=np.int64)
        gq = np.array([sorted(rng.choice(N, q, replace=False)) for _ in range(11)], dtype=np.int64)
        A = rng.randn(23, k, k); ick = (A @ A.transpose(0, 2, 1) + np.eye(k)).ravel()
        B = rng.randn(11, q, q); icq = (B @ B.transpose(0, 2, 1) + np.eye(q)).ravel()
        mk = 10 * rng.rand(23); mki = rng.randint(0, 50, size=23).astype(np.int64)
        P = rng.randn(N, N); P = P @ P.T
        psi = np.zeros((N * N, 23)); cm.assemble_psi_c(psi.ravel(), N, k, 23, gk.ravel(), ick)
        PHI = np.zeros(N * N); cm.objectiveK_c(PHI, N, k, 23, mk, gk.ravel(), ick)
        PHIi = np.zeros(N * N); cm.objectiveK_c(PHIi, N, k, 23, mki, gk.ravel(), ick)
        grad = np.zeros(23); cm.gradK_c(grad, k, 23, gk.ravel(), ick, P[0])
        X = np.zeros((N, 23)); cm.cleanupK_c(X.ravel(), k, 23, gk.ravel(), ick, P[0])
        hess = np.zeros((23, 11)); cm.hessKQ_c(hess.ravel(), N, k, q, 23, 11, gk.ravel(), gq.ravel(), ick, icq, P.ravel())
        tag = "k%dq%d_" % (k, q)
        for name, val in (("gk", gk), ("gq", gq), ("ick", ick), ("icq", icq), ("mk", mk), ("mki", mki), ("P", P),
                          ("psi", psi), ("PHI", PHI), ("PHIi", PHIi), ("grad", grad), ("X", X), ("hess", hess)):
            out[tag + name] = val
    np.savez_compressed(os.path.join(OUT, "cmisc_known_answers.npz"), **out)
    print("cmisc_known_answers.npz")


def edge_ms(sap, m, rng):
    """edge-case allocations the reference handles specially (misc.py:464,467-470)"""
    L = sap.L
    cases = {"base": m.copy()}
    mz = m.copy(); mz[rng.rand(L) < 0.7] = 0.0; mz[0] = 3.0          # sparse but model 0 sampled
    cases["sparse"] = mz
    # drop the LAST model everywhere: every group containing model N-1 gets m=0
    md = m.copy()
    last = np.concatenate([(g == sap.N - 1).any(axis=1) for g in sap.groups])
    md[last] = 0.0
    cases["drop_last_model"] = md
    # only groups inside {0,1,2} sampled (several models unsampled)
    mo = np.zeros(L)
    inside = np.concatenate([(g <= 2).all(axis=1) for g in sap.groups])
    mo[inside] = m[inside]
    cases["only_first3"] = mo
    cases["tiny"] = 0.01 * np.ones(L)                                  # -> inf (misc.py:464)
    mi = np.floor(m).astype(np.int64); mi[0] = max(mi[0], 1)
    cases["int64"] = mi                                                # integer allocation (after projection)
    return cases


def gen_sap(bluest, misc, n, kmax, n_out, fname, store_grad_outputs=None, with_psi=False, with_hess=False):
    prob = synth.problem(n, kmax, n_out)
    out = {"n": n, "kmax": kmax, "n_out": n_out}
    for o in range(n_out):
        sap = bluest.SAP(prob["C"][o].copy(), kmax, lists_of(prob["groups"]), prob["costs"], verbose=False)
        rng = np.random.RandomState(99 + o)
        cases = edge_ms(sap, prob["m"][o], rng)
        if o == 0:
            ic = np.concatenate(sap.invcovs)
            if len(ic) <= 200000:
                out["invcovs_o0"] = ic
            else:                                   # keep the fixture small: strided subset + moments
                out["invcovs_o0_sub"] = ic[::101]; out["invcovs_o0_sum"] = ic.sum(); out["invcovs_o0_norm"] = np.linalg.norm(ic)
            if with_psi:
                out["psi_o0"] = sap.psi
        for name, m in cases.items():
            tag = "o%d_%s_" % (o, name)
            mf = m.astype(np.float64)
            if name in ("sparse", "drop_last_model", "only_first3", "int64"):
                out[tag + "m"] = m
            for delta in ((0.0, 1.0e-6) if name in ("base", "drop_last_model") else (0.0,)):
                dtag = tag + ("d%g_" % delta if delta else "")
                V = sap.variance(mf, delta=delta)
                out[dtag + "V"] = V
                if name == "tiny":
                    Vg, g = misc.variance_GH_full(mf, sap.psi, sap.groups, sap.sizes, sap.invcovs, delta=delta, nohess=True)[:2]
                    out[dtag + "Vgh"] = Vg; out[dtag + "grad_isinf"] = np.isinf(g).all()
                    continue
                Vgh, grad, hess = sap.variance_GH(mf, delta=delta, nohess=not (with_hess and name == "base" and delta == 0.0))
                out[dtag + "Vgh"] = Vgh
                out[dtag + "PHI"] = sap.get_phi(mf, delta=delta)
                if hess is not None:
                    out[dtag + "hess"] = hess
                if store_grad_outputs is None or (o in store_grad_outputs and name == "base" and delta == 0.0):
                    out[dtag + "grad"] = grad
                else:
                    out[dtag + "grad_sub"] = grad[::97]
                    out[dtag + "grad_norm"] = np.linalg.norm(grad)
                    out[dtag + "grad_sum"] = grad.sum()
        # int64 allocation through the native overload (cmisc.cpp:105) directly
        import _cmisc_bluest as cm
        PHI = np.zeros(n * n)
        mi = cases["int64"]
        for k in range(1, kmax + 1):
            cm.objectiveK_c(PHI, n, k, sap.sizes[k], np.ascontiguousarray(mi[sap.cumsizes[k - 1]:sap.cumsizes[k]]),
                            sap.groups[k - 1].ravel(), sap.invcovs[k - 1])
        out["o%d_int64_PHI_native" % o] = PHI.reshape(n, n)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "%.1f KB" % (os.path.getsize(os.path.join(OUT, fname)) / 1024))


def gen_estimator(bluest, fname):
    """compute_BLUE_estimator (sap.py:99-119 + misc.py:518-544) on the n = 6 all-groups problem: integer samples with most
    groups unsampled, scalar sums per (group, model)"""
    n, kmax = 6, 6
    prob = synth.problem(n, kmax, 1)
    sap = bluest.SAP(prob["C"][0].copy(), kmax, lists_of(prob["groups"]), prob["costs"], verbose=False)
    rng = np.random.RandomState(21)
    out = {"n": n, "kmax": kmax}
    for case in range(3):
        samples = np.zeros(sap.L, dtype=np.int64)
        pick = rng.choice(sap.L, 9, replace=False)
        samples[pick] = rng.randint(1, 40, size=9)
        samples[0] = 3 + case                                  # the group {0}: model 0 is sampled
        if case == 2:
            samples[1] = 0; samples[2] = 0                     # some models never sampled -> restricted system (misc.py:523-525)
        sums, flat = [], []
        for k in range(1, kmax + 1):
            for i in range(sap.sizes[k]):
                li = sap.cumsizes[k - 1] + i
                v = samples[li] * (1.0 + 0.1 * rng.randn(k)) if samples[li] > 0 else np.zeros(k)
                sums.append(list(v)); flat.append(v)
        mu, var = sap.compute_BLUE_estimator(sums, samples=samples)
        out["samples%d" % case] = samples
        out["sums%d" % case] = np.concatenate(flat)
        out["mu%d" % case] = mu
        out["var%d" % case] = var
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname)


def gen_mosap(bluest, fname):
    """multi-output with DIFFERENT group sets per output (non-identity mappings, mosap.py:54-67)"""
    n, n_out, kmax = 6, 3, 3
    prob = synth.problem(n, kmax, n_out)
    allg = prob["groups"]
    rng = np.random.RandomState(5)
    multi_groups = []
    for o in range(n_out):
        mg = []
        for k in range(kmax):
            keep = rng.rand(len(allg[k])) < 0.7
            keep[0] = True                                  # keep a group with model 0 for every size
            mg.append(allg[k][keep])
        multi_groups.append(mg)
    # union, sorted, as blue_models.py:493-501
    groups = []
    for k in range(kmax):
        s = sorted({tuple(map(int, g)) for o in range(n_out) for g in multi_groups[o][k]})
        groups.append(np.array(s, dtype=np.int64))
    w = prob["w"]
    costs = synth.group_costs(groups, w)
    multi_costs = [synth.group_costs(mg, w) for mg in multi_groups]
    mos = bluest.MOSAP([c.copy() for c in prob["C"]], kmax, [kmax] * n_out, lists_of(groups),
                       [lists_of(mg) for mg in multi_groups], costs, multi_costs, verbose=False)
    L = mos.L
    m = 10 * np.random.RandomState(11).rand(L)
    Vs = mos.variances(m)
    Vgh, grads, _ = mos.variance_GH(m, nohess=True)
    out = {"n": n, "n_out": n_out, "kmax": kmax, "m": m, "Vs": np.array(Vs), "Vgh": np.array(Vgh), "e": mos.e}
    # sparsification of an allocation without changing variance or cost (mosap.py:102-111, 125-210)
    out["X_cleanup"] = mos.get_cleanup_matrices(m)
    mc = mos.cleanup_solution(m.copy())
    out["m_clean"] = mc
    out["V_clean"] = np.array(mos.variances(mc))
    out["costs"] = costs
    for o in range(n_out):
        out["map%d" % o] = mos.mappings[o]
        out["grad%d" % o] = grads[o]
        for k in range(kmax):
            out["mg%d_k%d" % (o, k + 1)] = np.asarray(multi_groups[o][k])
    for k in range(kmax):
        out["g_k%d" % (k + 1)] = groups[k]
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname)


def gen_hh(bluest, fname):
    """paper known answer: Hodgkin-Huxley model graph + stored allocation, K=7"""
    d = dict(np.load(os.path.join(REF, "examples/paper_examples/hodgkin-huxley/model_graph_data.npz")))
    samples = np.load(os.path.join(REF, "examples/paper_examples/hodgkin-huxley/samples.npz"))["samples"]
    n = int(d["M"]); n_out = int(d["n_outputs"]); kmax = 7
    Cs = [d["C%d" % o] for o in range(n_out)]
    assert all(np.isfinite(C).all() for C in Cs)
    groups = synth.all_groups(n, kmax)
    costs = synth.group_costs(groups, d["costs"])
    mos = bluest.MOSAP([C.copy() for C in Cs], kmax, [kmax] * n_out, lists_of(groups),
                       [lists_of(groups) for _ in range(n_out)], costs, [costs] * n_out, verbose=False)
    assert mos.L == len(samples)
    Vs = np.array(mos.variances(samples))
    Vgh, grads, _ = mos.variance_GH(samples.astype(np.float64), nohess=True)
    eps = np.sqrt(np.array([C[0, 0] for C in Cs])) / 1000          # blue_hodgkin-huxley.py:419
    out = {"n": n, "n_out": n_out, "kmax": kmax, "costs": d["costs"], "samples": samples, "Vs": Vs,
           "Vgh": np.array(Vgh), "errors_over_eps": np.sqrt(Vs) / eps, "total_cost": float(samples @ costs)}
    for o in range(n_out):
        out["C%d" % o] = Cs[o]
        out["grad%d_sub" % o] = grads[o][::13]
        out["grad%d_norm" % o] = np.linalg.norm(grads[o])
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, out["errors_over_eps"], out["total_cost"])


def gen_ns(bluest, fname):
    """second paper data set: Navier-Stokes, 12 models, 6 outputs, K = 7 (bluest_NS.py:142), covariances
    with cond up to 1.5e11.  The stored model graphs are complete and SG lists all 12 models for every output, so the groups of
    `setup_solver(K=7)` (blue_models.py:462-476) are all 3301 subsets of up to 7 models for every output.  Two cases:
      full    every output uses every group (identity mappings);
      ragged  every output keeps a seeded ~60 % of the groups of each size (plus the first group of each size): the union /
              mapping logic of blue_models.py:491-501 and mosap.py:54-67 on real covariances.
    Allocations: a dense seeded one and a sparse one (40 groups + {0}); outputs of MOSAP.variances / variance_GH."""
    d = dict(np.load(os.path.join(REF, "examples/paper_examples/navier_stokes/NS_model_data_full.npz")))
    n = int(d["M"]); n_out = int(d["n_outputs"]); kmax = 7
    assert np.array_equal(d["SG"], np.tile(np.arange(n), (n_out, 1)))
    Cs = [d["C%d" % o] for o in range(n_out)]
    assert all(np.isfinite(C).all() and (C != 0).all() for C in Cs)          # complete graphs: every subset is a clique
    allg = synth.all_groups(n, kmax)
    out = {"n": n, "n_out": n_out, "kmax": kmax, "costs": d["costs"], "eps": 1e-3 * np.sqrt(np.array([C[0, 0] for C in Cs]))}   # bluest_NS.py:115
    for o in range(n_out):
        out["C%d" % o] = Cs[o]
    rng = np.random.RandomState(2026)
    for case in ("full", "ragged"):
        if case == "full":
            multi_groups = [[g.copy() for g in allg] for _ in range(n_out)]
        else:
            multi_groups = []
            for o in range(n_out):
                mg = []
                for k in range(kmax):
                    keep = rng.rand(len(allg[k])) < 0.6
                    keep[0] = True                                  # keeps a group with model 0 for every size
                    mg.append(allg[k][keep])
                multi_groups.append(mg)
        groups = []
        for k in range(kmax):                                       # union, sorted, as blue_models.py:491-501
            rows = sorted({tuple(map(int, g)) for o in range(n_out) for g in multi_groups[o][k]})
            groups.append(np.array(rows, dtype=np.int64).reshape(-1, k + 1))
        costs = synth.group_costs(groups, d["costs"])
        multi_costs = [synth.group_costs(mg, d["costs"]) for mg in multi_groups]
        mos = bluest.MOSAP([C.copy() for C in Cs], kmax, [kmax] * n_out, lists_of(groups),
                           [lists_of(mg) for mg in multi_groups], costs, multi_costs, verbose=False)
        L = mos.L
        m_dense = 10 * rng.rand(L)
        m_sparse = np.zeros(L)
        pick = rng.choice(L, 40, replace=False)
        m_sparse[pick] = 1 + 50 * rng.rand(40)
        m_sparse[0] = 7.0                                           # group {0}: model 0 of every output is sampled
        out[case + "_L"] = L
        out[case + "_group_costs"] = costs
        for k in range(kmax):
            out["%s_g_k%d" % (case, k + 1)] = groups[k]
        for o in range(n_out):
            out["%s_map%d" % (case, o)] = np.asarray(mos.mappings[o], dtype=np.int64)
        for tag, m in (("dense", m_dense), ("sparse", m_sparse)):
            Vs = np.array(mos.variances(m))
            Vgh, grads, _ = mos.variance_GH(m, nohess=True)
            out["%s_m_%s" % (case, tag)] = m
            out["%s_Vs_%s" % (case, tag)] = Vs
            out["%s_Vgh_%s" % (case, tag)] = np.array(Vgh)
            for o in range(n_out):
                out["%s_grad%d_%s_sub" % (case, o, tag)] = grads[o][::7]
                out["%s_grad%d_%s_norm" % (case, o, tag)] = np.linalg.norm(grads[o])
        if case == "full":
            out["cond_phi_dense"] = np.array([np.linalg.cond(mos.SAPS[o].get_phi(m_dense)) for o in range(n_out)])
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, out["full_Vs_dense"], out["ragged_L"], out["cond_phi_dense"])


def gen_spg(bluest, spgmod, n, kmax, fname, maxit):
    """reference spg() (spg.py:39-132) driven by the reference's variance / variance_GH callbacks in the scaled
    variable x = cost*m/B, with the build-defined simplex projection.  Records EVERY callback evaluation so the
    build's driver can be compared call by call (trajectory parity defines m* parity)."""
    prob = synth.problem(n, kmax, 1)
    sap = bluest.SAP(prob["C"][0].copy(), kmax, lists_of(prob["groups"]), prob["costs"], verbose=False)
    w = prob["costs"]; B = prob["budget"]
    scale = B / w
    fvals, gnorms = [], []

    def feval(x):
        try:
            f = sap.variance(scale * x)
        except AssertionError:                      # model 0 dropped out of the trial point (misc.py:470)
            f = np.inf
        fvals.append(f)
        return f

    def geval(x):
        g = scale * sap.variance_GH(scale * x, nohess=True)[1]
        gnorms.append(np.linalg.norm(g))
        return g

    x0 = np.ones(sap.L) / sap.L
    res = spgmod.spg(feval, geval, orc.simplex_projection, x0, eps=1.0e-9, maxit=maxit, max_fevals=10 ** 5,
                     verbose=False)
    out = {"n": n, "kmax": kmax, "maxit": maxit, "eps": 1.0e-9, "fvals": np.array(fvals), "gnorms": np.array(gnorms),
           "x": res["x"], "f": res["f"], "gpmax": res["gpmax"], "it": res["it"], "count": res["count"],
           "solver_info": res["solver_info"]}
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "it", res["it"], "count", res["count"], "f", res["f"], "gpmax", res["gpmax"], "info", res["solver_info"])


def gen_intproj(bluest, misc, fname):
    """integer projection known answers (misc.py:177-382): continuous allocations -> reference's best integer point"""
    out = {}
    # single output, n=6 all groups: continuous solution = end point of the recorded SPG run
    G = np.load(os.path.join(OUT, "spg_traj_n6.npz"))
    prob = synth.problem(6, 6, 1)
    sap = bluest.SAP(prob["C"][0].copy(), 6, lists_of(prob["groups"]), prob["costs"], verbose=False)
    B = prob["budget"]
    sol = B * G["x"] / prob["costs"]
    out["s_sol"] = sol
    val, fval = misc.best_closest_integer_solution_BLUE(sol, sap.psi, prob["costs"], sap.e, budget=B)
    out["s_budget_val"] = val; out["s_budget_fval"] = fval
    eps = np.sqrt(sap.variance(sol))
    val, fval = misc.best_closest_integer_solution_BLUE(sol, sap.psi, prob["costs"], sap.e, eps=eps)
    out["s_eps"] = eps; out["s_eps_val"] = val; out["s_eps_fval"] = fval
    lb, ub, idx = misc.get_feasible_integer_bounds(sol, 6, e=sap.e)
    out["s_lb"] = lb; out["s_ub"] = ub; out["s_idx"] = idx
    # multi output, n=7, k<=3, 3 outputs, a hand-made sparse continuous point with fractional entries
    n, kmax, n_out = 7, 3, 3
    prob = synth.problem(n, kmax, n_out)
    groups = prob["groups"]
    mos = bluest.MOSAP([c.copy() for c in prob["C"]], kmax, [kmax] * n_out, lists_of(groups), [lists_of(groups) for _ in range(n_out)],
                       prob["costs"], [prob["costs"]] * n_out, verbose=False)
    rng = np.random.RandomState(17)
    sol = np.zeros(mos.L)
    pick = rng.choice(mos.L, 9, replace=False)
    sol[pick] = 0.3 + 40 * rng.rand(9)
    sol[0] = 5.7                         # model 0 alone
    sol[mos.L - 1] = 12.4
    out["m_sol"] = sol
    psis = [mos.SAPS[o].psi for o in range(n_out)]
    B = float(sol @ prob["costs"]) * 1.0005
    val, fval = misc.best_closest_integer_solution_BLUE_multi(sol, psis, prob["costs"], mos.e, mos.mappings, budget=B)
    out["m_budget"] = B; out["m_budget_val"] = val; out["m_budget_fval"] = fval
    eps = np.sqrt(np.array(mos.variances(sol))) * 1.02
    val, fval = misc.best_closest_integer_solution_BLUE_multi(sol, psis, prob["costs"], mos.e, mos.mappings, eps=eps)
    out["m_eps"] = eps; out["m_eps_val"] = val; out["m_eps_fval"] = fval
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, out["s_budget_fval"], out["s_eps_fval"], out["m_budget_fval"], out["m_eps_fval"], len(out["s_idx"]))


def gen_singular(bluest, fname):
    """a rank-deficient information matrix with EVERY model touched: models 2 and 3 are perfectly correlated, so the covariance
    blocks of the groups that contain both are singular and their pseudo-inverses (sap.py:74) leave Phi singular on span(e2 - e3).
    Records what the reference returns there: `variance` goes through np.linalg.solve (misc.py:472), `variance_GH` through
    np.linalg.pinv (misc.py:487,490) -- the two disagree with each other (the build reports BLUEST_EVAL_SINGULAR instead)."""
    rng = np.random.RandomState(5)
    n = 5
    Z = rng.randn(n, 12)
    C = Z @ Z.T / 12
    C[3, :] = C[2, :]; C[:, 3] = C[:, 2]; C[3, 3] = C[2, 2]
    groups = [[[0], [1], [4]], [[0, 1], [2, 3], [1, 4]], [[0, 2, 3], [1, 2, 3]]]
    costs = np.ones(8)
    sap = bluest.SAP(C.copy(), 3, [[list(g) for g in gk] for gk in groups], costs, verbose=False)
    m = np.array([3.0, 2.0, 1.5, 4.0, 2.5, 1.0, 3.5, 2.0])
    PHI = sap.get_phi(m)
    try:
        v_solve, asserted = sap.variance(m), 0
    except AssertionError:
        v_solve, asserted = np.nan, 1
    V, g, _ = sap.variance_GH(m, nohess=True)
    out = {"n": n, "C": C, "m": m, "PHI": PHI, "eigmin": np.linalg.eigvalsh(PHI)[0], "variance_solve": v_solve,
           "variance_asserted": asserted, "Vgh_pinv": V, "grad_pinv": g, "invcovs": np.concatenate(sap.invcovs)}
    for k, gk in enumerate(groups):
        out["g_k%d" % (k + 1)] = np.array(gk, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, "eigmin", out["eigmin"], "variance (solve)", v_solve, "variance_GH (pinv)", V)


def gen_spg_bound(bluest, spgmod, n, kmax, fname, maxit):
    """best objective the reference's plain spg() (spg.py:39-132, its own defaults) reaches on the sample-allocation problem
    with the reference's callbacks within `maxit` iterations: an upper bound the build's solver has to beat"""
    prob = synth.problem(n, kmax, 1)
    sap = bluest.SAP(prob["C"][0].copy(), kmax, lists_of(prob["groups"]), prob["costs"], verbose=False)
    scale = prob["budget"] / prob["costs"]
    best = [np.inf]

    def feval(x):
        try:
            f = sap.variance(scale * x)
        except AssertionError:
            f = np.inf
        best[0] = min(best[0], f)
        return f

    def geval(x):
        return scale * sap.variance_GH(scale * x, nohess=True)[1]

    res = spgmod.spg(feval, geval, orc.simplex_projection, np.ones(sap.L) / sap.L, eps=1.0e-9, maxit=maxit, max_fevals=10 ** 5, verbose=False)
    np.savez_compressed(os.path.join(OUT, fname), n=n, kmax=kmax, maxit=maxit, best_f=best[0], it=res["it"], count=res["count"],
                        gpmax=res["gpmax"], solver_info=res["solver_info"])
    print(fname, "best f", best[0], "it", res["it"], "count", res["count"], "gpmax", res["gpmax"])


def main():
    os.makedirs(OUT, exist_ok=True)
    cm, bluest, misc, spgmod = import_reference()
    gen_cmisc(cm)
    gen_sap(bluest, misc, 5, 5, 1, "sap_n5_all.npz", with_psi=True, with_hess=True)
    gen_sap(bluest, misc, 6, 6, 2, "sap_n6_all.npz", with_psi=True)
    gen_sap(bluest, misc, 12, 12, 1, "sap_n12_all.npz")
    gen_sap(bluest, misc, 20, 5, 8, "sap_n20_k5_o8.npz", store_grad_outputs=(0,))
    gen_mosap(bluest, "mosap_n6_o3_ragged.npz")
    gen_estimator(bluest, "estimator_n6_known_answers.npz")
    gen_hh(bluest, "hh_paper_known_answer.npz")
    gen_spg(bluest, spgmod, 6, 6, "spg_traj_n6.npz", maxit=60)
    gen_spg(bluest, spgmod, 12, 4, "spg_traj_n12_k4.npz", maxit=40)
    gen_intproj(bluest, misc, "intproj_known_answers.npz")
    gen_singular(bluest, "singular_phi_known_answer.npz")
    gen_spg_bound(bluest, spgmod, 12, 12, "spg_bound_n12_all.npz", maxit=400)
    gen_ns(bluest, "ns_paper_known_answer.npz")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--new-only":       # the fixtures added in round 3, leaving the others untouched
        cm_, bluest_, misc_, spgmod_ = import_reference()
        gen_ns(bluest_, "ns_paper_known_answer.npz")
    else:
        main()
