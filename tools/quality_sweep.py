"""Default solve vs a much more patient one (tight full-problem stages, more pricing rounds, long stall windows) on a range of
problem shapes: how far above the patient answer does the default end?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
cfgs = [(16, 5, 1), (16, 5, 2), (16, 5, 4), (18, 5, 1), (18, 5, 3), (20, 4, 2), (20, 4, 8), (22, 4, 1), (22, 4, 4), (22, 5, 2), (14, 6, 1), (14, 6, 3), (24, 4, 2)]
patient = {"polish_full_loose": 1.0, "polish_rounds": 6, "polish_stall_window": 150, "stall_window": 200, "maxit": 20000}
worst = 0.0
for n, k, o in cfgs:
    prob = synth.problem(n, k, o)
    g = prob["groups"]
    mos = MOSAP(prob["C"], k, [k] * o, [a.copy() for a in g], [[a.copy() for a in g] for _ in range(o)], prob["costs"], [prob["costs"]] * o, verbose=False)
    B = prob["budget"]
    res = []
    for sp in (None, patient):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = mos.solve(budget=B, solver="spg", continuous_relaxation=True, solver_params=sp)
        torch.cuda.synchronize()
        res.append((max(mos.variances(m)), (time.perf_counter() - t0) * 1e3, mos.solver_info["it"]))
    gap = res[0][0] / min(res[0][0], res[1][0]) - 1
    worst = max(worst, gap)
    print("n=%d k=%d o=%d K_tot=%d: default V %.9e (%.0f ms, %d it)  patient V %.9e (%.0f ms)  default above best by %.1e" %
          (n, k, o, mos.L, res[0][0], res[0][1], res[0][2], res[1][0], res[1][1], gap), flush=True)
print("worst gap of the default: %.2e" % worst)
