"""Experiment: smoothing-exponent schedules of the multi-output SPG on a few problems: true objective max_o V_o, iterations, seconds."""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

inf = float("inf")
schedules = {"32,512": (32, 512), "32,inf": (32, inf), "32,512,inf": (32, 512, inf), "32,2048,inf": (32, 2048, inf), "64,2048": (64, 2048)}
problems = [(20, 5, 8), (20, 5, 3), (15, 4, 5), (12, 6, 4), (10, 10, 6)]
for (n, kmax, n_out) in problems:
    prob = synth.problem(n, kmax, n_out)
    groups = prob["groups"]
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                prob["costs"], [prob["costs"]] * n_out, verbose=False)
    mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
    for name, sched in schedules.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, solver_params={"smoothing_p": sched})
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("n=%d k=%d o=%d  p=%-12s maxV %.9g  it %5d evals %5d  %.3f s  nnz %d" % (
            n, kmax, n_out, name, max(mos.variances(m)), mos.solver_info["it"], mos.solver_info["count"], dt, int((m > 0).sum())), flush=True)
