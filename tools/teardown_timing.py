"""Experiment: what does releasing a solve's hipGraphs cost, and where does it land?  Solves the headline problem three times and
times (a) the explicit release of the solver's graphs right after the solve, (b) the next blocking copy, (c) the next set-up."""
import gc
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bluest_amd import spg_device, synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

n, kmax, n_out = 20, 5, 8
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
torch.zeros(1, device="cuda")
explicit = "--explicit" in sys.argv
solvers = []
orig_init = spg_device.DeviceSpg.__init__


def init(self, *a, **k):
    orig_init(self, *a, **k)
    solvers.append(self)


spg_device.DeviceSpg.__init__ = init
for rep in range(4):
    gc.collect()
    gc.disable()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                prob["costs"], [prob["costs"]] * n_out, verbose=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n_graphs = sum(len(gs) for s in solvers for gs in s.graph_sets.values())
    if explicit:
        for s in solvers:
            s.graph_sets = {}
            s.graphs = None
    del solvers[:]
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    torch.zeros(1, device="cuda").cpu()
    t4 = time.perf_counter()
    mos = None
    gc.collect()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    torch.zeros(1, device="cuda").cpu()
    t6 = time.perf_counter()
    gc.enable()
    print("rep %d: setup %.1f ms, solve %.1f ms, %d graphs, release graphs %.1f ms, blocking copy %.1f ms, drop plan %.1f ms, blocking copy %.1f ms"
          % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n_graphs, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3), flush=True)
