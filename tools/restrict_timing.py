"""How long does the working set's restricted plan take to build (headline problem, 16*N groups)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
n, k, o = 20, 5, 8
prob = synth.problem(n, k, o)
g = prob["groups"]
mos = MOSAP(prob["C"], k, [k] * o, [a.copy() for a in g], [[a.copy() for a in g] for _ in range(o)], prob["costs"], [prob["costs"]] * o, verbose=False)
rng = np.random.RandomState(0)
keep = np.unique(np.concatenate([np.flatnonzero(mos.e > 0)[:40], rng.choice(mos.L, 300, replace=False)]))
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sub = mos._restricted_plan(keep)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("restricted plan of %d groups: %.2f ms" % (len(keep), (t1 - t0) * 1e3), flush=True)
    del sub
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
sub = mos._restricted_plan(keep); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
