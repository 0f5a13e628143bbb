"""The default solve on OTHER covariances than the ones every record of this repository was taken on: the headline shape (and two
others) with the Wishart seeds shifted, and with the reference's self-test style C = G^T G (bluest/sap.py:463) -- certified gap, time,
and which method answered.      python tools/seed_sweep.py [key=value ...]     (overrides for the second-order finish, e.g. enter_per_round=40)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

prm = {kv.split("=")[0]: eval(kv.split("=", 1)[1]) for kv in sys.argv[1:]}
shapes = prm.pop("shapes", ((20, 5, 8), (16, 4, 3), (25, 6, 1), (12, 12, 1)))
print("# overrides: %s" % (prm or "none"))
worst, total = 0.0, 0.0
for n, k, o in shapes:
    for shift in range(0, 80, 10):
        prob = synth.problem(n, k, o)
        for q in range(o):
            if shift < 60:
                prob["C"][q] = synth.wishart_covariance(n, 100 + shift + q)[0]
            else:                                                    # the reference's self-test style: C = G^T G, no decay
                G = np.random.RandomState(500 + shift + q).randn(n, n)
                prob["C"][q] = G.T @ G
        g = prob["groups"]
        mos = MOSAP(prob["C"], k, [k] * o, [a.copy() for a in g], [[a.copy() for a in g] for _ in range(o)], prob["costs"], [prob["costs"]] * o, verbose=False)
        best = np.inf
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, solver_params={"newton": dict(prm)} if prm else None)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        si = mos.solver_info
        gap = si.get("certified_gap", np.nan)
        worst = max(worst, gap if np.isfinite(gap) else 1.0)
        total += best
        print("n=%2d k=%2d o=%d %s seed+%2d: %-6s gap %9.2e  %6.2f ms  rounds %3s  it %4s  cond(C) %.1e" % (n, k, o, "Wishart" if shift < 60 else "G^T G  ", shift, si.get("method"), gap, best * 1e3,
              si.get("rounds"), si.get("it"), max(np.linalg.cond(c) for c in prob["C"])), flush=True)
print("worst certified gap: %.2e; sum of the warm solves %.1f ms" % (worst, total * 1e3))
