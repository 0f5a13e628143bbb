"""Sweep of polish_full_loose (stall tolerance of the full-problem stages before the working set is chosen): objective and time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
cfgs = [(16, 5, 2), (20, 5, 8), (20, 5, 1), (25, 6, 1), (14, 6, 3)]
for n, k, o in cfgs:
    prob = synth.problem(n, k, o)
    g = prob["groups"]
    mos = MOSAP(prob["C"], k, [k] * o, [a.copy() for a in g], [[a.copy() for a in g] for _ in range(o)], prob["costs"],
                [prob["costs"]] * o, verbose=False)
    B = prob["budget"]
    rows = []
    for sp in ({"price_interior": 0.0}, {"price_interior": 1e-4}, {"price_interior": 1e-3}, {"price_interior": 1e-2},
               {"price_interior": 1e-3, "polish_rounds": 4}, {"polish_full_loose": 5.0, "price_interior": 1e-3}):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m = mos.solve(budget=B, solver="spg", continuous_relaxation=True, solver_params=sp)
            torch.cuda.synchronize(); t1 = time.perf_counter()
        rows.append((sp, (t1 - t0) * 1e3, max(mos.variances(m)), mos.solver_info["it"], mos.solver_info["count"]))
    best = min(r[2] for r in rows)
    for sp, ms, V, it, cnt in rows:
        print("n=%d k=%d o=%d L=%d %-50s %6.1f ms  V %.9e  (+%.1e)  it %d evals %d" % (n, k, o, mos.L, sp, ms, V, V / best - 1, it, cnt), flush=True)
