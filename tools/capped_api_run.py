"""max_model_samples through the product API (MOSAP.solve) on a synthetic problem: python tools/capped_api_run.py n kmax n_out ncaps
the ncaps most sampled models are capped at half of what the free optimum gives them; prints time, method, certified gap, cap usage"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

n, kmax, n_out, ncaps = (int(a) for a in sys.argv[1:5])
prob = synth.problem(n, kmax, n_out)
groups, w, B = prob["groups"], prob["costs"], prob["budget"]
mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)], w, [w] * n_out, verbose=False)
m_free = mos.solve(budget=B, solver="spg", continuous_relaxation=True)
usage = np.array([float(mos.ES[i] @ m_free) for i in range(n)])
caps = np.full(n, np.inf)
for i in np.argsort(-usage)[:ncaps]:
    caps[i] = max(1.0, np.floor(0.5 * usage[i]))
for rep in range(2):
    t0 = time.perf_counter()
    m = mos.solve(budget=B, solver="spg", continuous_relaxation=True, max_model_samples=caps)
    dt = time.perf_counter() - t0
    si = mos.solver_info
    used = [float(mos.ES[i] @ m) / caps[i] for i in range(n) if np.isfinite(caps[i])]
    print("n=%d k=%d o=%d caps=%d rep %d: %.3f s  method %s (%s)  gap %s  max V %.10e (free %.10e)  usage %s  budget %.12f" % (
        n, kmax, n_out, ncaps, rep, dt, si.get("method"), si.get("caps", "-"), si.get("certified_gap"), max(mos.variances(m)), max(mos.variances(m_free)),
        np.array2string(np.array(used), precision=6), float(m @ w) / B))
