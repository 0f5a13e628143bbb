"""Experiment: re-solve the allocation problem restricted to the support selected at the end of the SPG run."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
n, kmax, n_out = (int(a) for a in (sys.argv[1:4] or (20, 5, 8)))
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
            prob["costs"], [prob["costs"]] * n_out, verbose=False)
for S_mult in (None, 4, 8, 16):
    params = {} if S_mult is None else {"sparsify_tol": 0.0}
    t0 = time.perf_counter()
    m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, solver_params=params or None)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("full solve %.3f s: nnz %d, max V %.9g" % (t1 - t0, (m > 0).sum(), max(mos.variances(m))))
    if S_mult is None:
        continue
    S = S_mult * n
    keep = np.sort(np.argsort(-prob["costs"] * m)[:S])
    cum = np.concatenate([[0], np.cumsum([len(g) for g in groups])])
    sub = [groups[k][keep[(keep >= cum[k]) & (keep < cum[k + 1])] - cum[k]] for k in range(kmax)]
    t0 = time.perf_counter()
    mos2 = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in sub], [[g.copy() for g in sub] for _ in range(n_out)],
                 prob["costs"][keep], [prob["costs"][keep]] * n_out, verbose=False)
    x0 = m[keep] * prob["budget"] / (prob["costs"][keep] @ m[keep])
    m2 = mos2.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, x0=x0, solver_params={"smoothing_p": (float("inf"),)})
    torch.cuda.synchronize(); t1 = time.perf_counter()
    mm = np.zeros_like(m); mm[keep] = m2
    print("  restricted to %3d groups: %.3f s (set-up included), %d iterations, nnz %d, max V %.9g (full operator: %.9g)" % (
        S, t1 - t0, mos2.solver_info["it"], (m2 > 0).sum(), max(mos2.variances(m2)), max(mos.variances(mm))))
