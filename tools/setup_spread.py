"""Experiment: spread of the warm MOSAP construction time when solves happen in between (who is slow when it is slow)."""
import cProfile
import gc
import pstats
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

n, kmax, n_out = (int(a) for a in (sys.argv[1:4] or (20, 5, 8)))
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
mos = None
worst = (0.0, None)
for rep in range(8):
    mos = None
    gc.collect()
    gc.disable()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                prob["costs"], [prob["costs"]] * n_out, verbose=False)
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t0
    gc.enable()
    print("construction %d: %.1f ms" % (rep, dt * 1e3), flush=True)
    if dt > worst[0]:
        worst = (dt, pr)
    if rep % 2 == 0:
        mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
print("slowest construction: %.1f ms" % (worst[0] * 1e3))
pstats.Stats(worst[1]).sort_stats("tottime").print_stats(8)
