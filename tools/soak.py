"""Soak: many solves on different problems; every result is feasible, finite and reproducible (the same problem solved twice
gives the same objective), the fused decision's ticket never leaves a solve hanging."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bluest_amd import synth
from bluest_amd.mosap import MOSAP

t_all = time.time()
for n, kmax, n_out in ((6, 3, 1), (8, 4, 3), (10, 3, 5), (12, 5, 2), (14, 4, 8), (16, 5, 1), (20, 4, 4), (20, 5, 8), (25, 5, 2)):
    prob = synth.problem(n, kmax, n_out)
    groups = prob["groups"]
    Fs = []
    for rep in range(2):
        mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                    prob["costs"], [prob["costs"]] * n_out, verbose=False)
        t0 = time.time()
        m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
        dt = time.time() - t0
        assert m is not None and np.isfinite(m).all() and (m >= 0).all() and abs(m @ prob["costs"] / prob["budget"] - 1) < 1e-9
        Fs.append(max(mos.variances(m)))
        eps = np.sqrt(np.array(mos.variances(m)) * 1.3)
        me = mos.solve(eps=eps, solver="spg", continuous_relaxation=True)
        assert me is not None and (np.array(mos.variances(me)) <= eps ** 2 * (1 + 1e-9)).all()
    assert abs(Fs[0] / Fs[1] - 1) < 1e-12, Fs
    print("n=%d k=%d o=%d: F %.6e, it %d, %.3f s" % (n, kmax, n_out, Fs[0], mos.solver_info["it"], dt), flush=True)
print("soak ok in %.1f s" % (time.time() - t_all))
