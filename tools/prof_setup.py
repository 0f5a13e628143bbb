import cProfile, pstats, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
n, kmax, n_out = 20, 5, 8
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
def mk():
    return MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                 prob["costs"], [prob["costs"]] * n_out, verbose=False)
mk()
pr = cProfile.Profile(); pr.enable(); t0=time.perf_counter(); mos = mk(); torch.cuda.synchronize(); t1=time.perf_counter(); pr.disable()
print("setup", t1-t0)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
