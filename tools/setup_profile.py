"""cProfile of the MOSAP constructor (headline problem): where the host side of the set-up goes."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
n, k, o = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (20, 5, 8)
prob = synth.problem(n, k, o)
g = prob["groups"]
def build():
    return MOSAP(prob["C"], k, [k] * o, [a.copy() for a in g], [[a.copy() for a in g] for _ in range(o)], prob["costs"], [prob["costs"]] * o, verbose=False)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); mos = build(); torch.cuda.synchronize()
    print("set-up %.2f ms" % ((time.perf_counter() - t0) * 1e3)); del mos
pr = cProfile.Profile(); pr.enable(); mos = build(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
