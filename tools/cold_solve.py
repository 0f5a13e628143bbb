"""Where the time of the FIRST solve of a process goes (library load, code-object load on first launch, graph captures)."""
import cProfile
import pstats
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
t00 = time.perf_counter()
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402
t0 = time.perf_counter()
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
t1 = time.perf_counter()
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402
n, kmax, n_out = (int(a) for a in (sys.argv[1:4] or (20, 5, 8)))
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
t2 = time.perf_counter()
mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
            prob["costs"], [prob["costs"]] * n_out, verbose=False)
torch.cuda.synchronize()
t3 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
torch.cuda.synchronize()
pr.disable()
t4 = time.perf_counter()
print("import torch %.2f s | first CUDA touch %.3f s | problem synthesis %.3f s | MOSAP (cold) %.3f s | solve (cold) %.3f s" % (
    t0 - t00, t1 - t0, t2 - t1, t3 - t2, t4 - t3))
pstats.Stats(pr).sort_stats("tottime").print_stats(10)
