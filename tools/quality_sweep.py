"""solver="spg" with the second-order finish over a range of problem shapes: certified gap, wall-clock, support size -- and the
first-order loop alone (method="spg") on the same problem for comparison.  python tools/quality_sweep.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
cfgs = [(5, 5, 1), (6, 3, 2), (8, 4, 3), (10, 3, 3), (12, 12, 1), (14, 6, 1), (14, 6, 3), (16, 5, 1), (16, 5, 2), (16, 5, 4), (18, 5, 1),
        (18, 5, 3), (20, 4, 2), (20, 4, 8), (20, 5, 1), (20, 5, 8), (22, 4, 1), (22, 4, 4), (22, 5, 2), (24, 4, 2), (25, 6, 1), (20, 3, 12),
        (30, 3, 2), (40, 2, 1)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    cfgs = cfgs[:8]
worst_gap, worst_rel = 0.0, 0.0
for n, k, o in cfgs:
    prob = synth.problem(n, k, o)
    g = prob["groups"]
    mos = MOSAP(prob["C"], k, [k] * o, [a.copy() for a in g], [[a.copy() for a in g] for _ in range(o)], prob["costs"], [prob["costs"]] * o, verbose=False)
    B = prob["budget"]
    res = []
    for sp in (None, None, {"method": "spg"}):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = mos.solve(budget=B, solver="spg", continuous_relaxation=True, solver_params=sp)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        res.append((max(mos.variances(m)), dt, dict(mos.solver_info), int((m > 0).sum())))
    V, dt, info, nnz = res[1]
    gap = info.get("certified_gap", float("nan"))
    rel = res[2][0] / V - 1
    worst_gap = max(worst_gap, gap if np.isfinite(gap) else 0.0)
    print("n=%2d k=%2d o=%2d K_tot=%6d: %-6s V %.10e  gap %.1e  %5.1f ms  rounds %2d newton %3d  nnz %2d | first-order alone: +%.1e, %5.1f ms" %
          (n, k, o, mos.L, info.get("method", "spg"), V, gap, dt, info.get("rounds", 0), info["it"], nnz, rel, res[2][1]), flush=True)
print("worst certified gap: %.2e" % worst_gap)
