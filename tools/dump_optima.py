"""Experiment helper: solve the BASELINE configurations on the GPU and store the continuous optima (for off-line analysis with the
CPU oracle's optimality certificate)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

out = {}
for n, kmax, n_out in ((20, 5, 1), (20, 5, 8), (25, 6, 1), (12, 12, 1)):
    prob = synth.problem(n, kmax, n_out)
    groups = prob["groups"]
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                prob["costs"], [prob["costs"]] * n_out, verbose=False)
    t0 = time.perf_counter()
    m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
    dt = time.perf_counter() - t0
    V = mos.variances(m)
    print(n, kmax, n_out, "maxV", max(V), "nnz", (m > 0).sum(), "time", dt, mos.solver_info, flush=True)
    out["m_%d_%d_%d" % (n, kmax, n_out)] = m
    out["V_%d_%d_%d" % (n, kmax, n_out)] = np.array(V)
    if n_out == 8:
        eps = np.array([np.sqrt(c[0, 0]) / 30.0 for c in prob["C"]])
        me = mos.solve(eps=eps, solver="spg", continuous_relaxation=True)
        out["meps_%d_%d_%d" % (n, kmax, n_out)] = me
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/r2_optima.npz", **out)
