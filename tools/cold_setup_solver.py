"""Cold (first call in the process) BLUEProblem.setup_solver() with the default integer projection: where the time goes."""
import cProfile
import pstats
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bluest_amd import BLUEProblem, synth  # noqa: E402

n, kmax, n_out = (int(a) for a in (sys.argv[1:4] or (20, 5, 8)))
prob = synth.problem(n, kmax, n_out)
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
p = BLUEProblem(n, C=[c.copy() for c in prob["C"]], costs=prob["w"], n_outputs=n_out, verbose=False)
out = p.setup_solver(K=kmax, budget=prob["budget"], solver="spg")
pr.disable()
print("cold BLUEProblem + setup_solver (integer): %.3f s, %d groups, cost %.3f" % (time.perf_counter() - t0, len(out["models"]), out["total_cost"]))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
