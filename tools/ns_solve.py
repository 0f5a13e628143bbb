"""Navier-Stokes paper problem end to end (eps mode and budget mode) on the GPU; dumps the allocations for an offline look."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bluest_amd import BLUEProblem  # noqa: E402

G = dict(np.load(os.path.join(ROOT, "tests", "golden", "ns_paper_known_answer.npz")))
n, n_out, kmax = int(G["n"]), int(G["n_out"]), int(G["kmax"])
Cs = [G["C%d" % o] for o in range(n_out)]
out = {}
for mode in ("eps", "budget"):
    p = BLUEProblem(n, C=[c.copy() for c in Cs], costs=G["costs"], n_outputs=n_out, verbose=False)
    t0 = time.perf_counter()
    kw = {"eps": list(G["eps"])} if mode == "eps" else {"budget": float(max(G["costs"]) * 1e4)}
    res = p.setup_solver(K=kmax, continuous_relaxation=True, **kw)
    dt = time.perf_counter() - t0
    m = p.MOSAP.samples
    V = np.array(p.MOSAP.variances(m))
    print(mode, "seconds", dt, "cost", res["total_cost"], "ratios", V / G["eps"] ** 2, "nnz", int((m > 0).sum()), p.MOSAP.solver_info)
    out[mode + "_m"] = m
    out[mode + "_mu"] = np.asarray(p.MOSAP.solver_info.get("multipliers", np.zeros(n_out)))
    out[mode + "_V"] = V
np.savez(os.path.join(ROOT, "gpurun_out", "ns_solve.npz"), **out)
