"""One MOSAP set-up + SPG solve of a synthetic problem (for profiling): python tools/one_solve.py [n k n_out [repeats]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
n, k, o = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (20, 5, 8)
prob = synth.problem(n, k, o)
for rep in range(int(sys.argv[4]) if len(sys.argv) > 4 else 2):
    t0 = time.perf_counter()
    mos = MOSAP(prob["C"], k, [k] * o, [g.copy() for g in prob["groups"]], [[g.copy() for g in prob["groups"]] for _ in range(o)],
                prob["costs"], [prob["costs"]] * o, verbose=False)
    m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
    torch.cuda.synchronize()
    print("rep %d: %.1f ms, %s" % (rep, (time.perf_counter() - t0) * 1e3, {k_: mos.solver_info.get(k_) for k_ in ("it", "count", "rounds", "certified_gap", "method")}), flush=True)
