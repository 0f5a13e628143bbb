"""Experiment: how the mass of an SPG allocation is distributed over its entries (headline problem)."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
n, kmax, n_out = (int(a) for a in (sys.argv[1:4] or (20, 5, 8)))
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
            prob["costs"], [prob["costs"]] * n_out, verbose=False)
m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
c = prob["costs"] * m
order = np.argsort(-c)
share = np.cumsum(c[order]) / c.sum()
print("nnz", (m > 0).sum(), "max V", max(mos.variances(m)))
for k in (5, 10, 15, 20, 30, 50, 100, 200, 400):
    if k <= len(order):
        print("top %3d entries hold %.8f of the budget; entry %d: m = %.3e samples (cost share %.2e)" % (k, share[k - 1], k, m[order[k - 1]], c[order[k - 1]] / c.sum()))
for S in (16, 32, 64, 128):
    keep = order[:S]
    mm = np.zeros_like(m); mm[keep] = m[keep]; mm *= c.sum() / (prob["costs"] @ mm)
    print("keep top %3d, rescale to the budget: max V %.9g" % (S, max(mos.variances(mm))))
