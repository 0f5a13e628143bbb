"""Warm solve time and iteration counts over a set of problem shapes (for A/B of a switch that perturbs the trajectory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
import json
SP = json.loads(os.environ["SOLVER_PARAMS"]) if os.environ.get("SOLVER_PARAMS") else None
cfgs = [(20, 5, 8), (20, 5, 1), (16, 5, 2), (18, 5, 3), (20, 4, 4), (22, 4, 2), (22, 5, 1), (14, 6, 3), (24, 4, 2), (18, 5, 1), (16, 5, 4), (22, 4, 8)]
if os.environ.get("MANY_SHAPES"):
    cfgs = [(n, k, o) for n in range(15, 25) for k in (4, 5) for o in (2, 3) if synth.n_groups(n, k) > 4500][:28]
tot_ms = tot_it = 0.0
for n, k, o in cfgs:
    prob = synth.problem(n, k, o)
    g = prob["groups"]
    mos = MOSAP(prob["C"], k, [k] * o, [a.copy() for a in g], [[a.copy() for a in g] for _ in range(o)], prob["costs"], [prob["costs"]] * o, verbose=False)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, solver_params=SP)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    tot_ms += best; tot_it += mos.solver_info["it"]
    print("n=%d k=%d o=%d K_tot=%d: %.1f ms, %d it, %d evals, V %.8e" % (n, k, o, mos.L, best, mos.solver_info["it"], mos.solver_info["count"], max(mos.variances(m))), flush=True)
print("TOTAL %.1f ms, %d iterations" % (tot_ms, tot_it))
