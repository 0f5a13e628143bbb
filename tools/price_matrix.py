"""Experiment: final objective of solver="spg" vs working-set pricing parameters, against the certified optima."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

LB = {(20, 5, 1): 6.7414518336e-04, (20, 5, 8): 9.4852263368e-04, (25, 6, 1): 5.7322290049e-04}
for (n, kmax, n_out), lb in LB.items():
    prob = synth.problem(n, kmax, n_out)
    groups = prob["groups"]
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                prob["costs"], [prob["costs"]] * n_out, verbose=False)
    for rounds in (2, 4, 8):
        for ptol in (1e-3, 1e-4, 1e-5):
            for sup in (8, 16):
                prm = {"polish_rounds": rounds, "price_tol": ptol, "polish_support": sup}
                mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, solver_params=prm)   # warm graphs
                t0 = time.perf_counter()
                m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, solver_params=prm)
                dt = time.perf_counter() - t0
                F = max(mos.variances(m))
                print("n=%d o=%d rounds=%d price_tol=%g support=%dN: gap %.3e  nnz %d  it %d  %.3f s" % (
                    n, n_out, rounds, ptol, sup, F / lb - 1, (m > 0).sum(), mos.solver_info["it"], dt), flush=True)
