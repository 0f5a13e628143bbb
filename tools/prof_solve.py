import cProfile, pstats, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
n, kmax, n_out = (int(a) for a in sys.argv[1:4])
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
            prob["costs"], [prob["costs"]] * n_out, verbose=False)
mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter()
mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
torch.cuda.synchronize(); t1 = time.perf_counter(); pr.disable()
print("solve", t1 - t0, mos.solver_info)
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
