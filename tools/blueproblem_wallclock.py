"""BLUEProblem.setup_solver() wall-clock on the headline problem (n=20, n_out=8, K=5) through the user-facing API."""
import sys, time, cProfile, pstats
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth, BLUEProblem
n, kmax, n_out = 20, 5, 8
prob = synth.problem(n, kmax, n_out)
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    p = BLUEProblem(n, C=[c.copy() for c in prob["C"]], costs=prob["w"], n_outputs=n_out, verbose=False)
    t1 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    out = p.setup_solver(K=kmax, budget=prob["budget"], solver="spg", continuous_relaxation=True)
    pr.disable()
    t2 = time.perf_counter()
    print("rep", rep, "BLUEProblem() %.3f s, setup_solver() %.3f s, errors max %.6e, cost %.6f, groups used %d" % (t1 - t0, t2 - t1, out["errors"].max(), out["total_cost"], len(out["models"])))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
