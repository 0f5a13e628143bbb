"""Experiment: SAP solve with and without the working-set polish on a few problems: objective, iterations, seconds."""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

problems = [(20, 5, 8), (25, 6, 1), (20, 5, 1), (25, 5, 4), (16, 8, 2), (20, 5, 3)]
for (n, kmax, n_out) in problems:
    prob = synth.problem(n, kmax, n_out)
    groups = prob["groups"]
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                prob["costs"], [prob["costs"]] * n_out, verbose=False)
    mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
    for name, prm in (("polish", {}), ("no polish", {"polish": False})):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, solver_params=prm)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("n=%d k=%d o=%d  %-10s maxV %.9g  it %5d evals %5d  %.3f s  nnz %d" % (
            n, kmax, n_out, name, max(mos.variances(m)), mos.solver_info["it"], mos.solver_info["count"], dt, int((m > 0).sum())), flush=True)
