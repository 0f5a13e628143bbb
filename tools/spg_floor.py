"""Experiment: cost floor of one device-SPG iteration graph.  (1) replay with every kernel predicated off (DONE = 1): pure
launch / dispatch cost of the node chain; (2) normal replays, host never looking; (3) the same iterations captured 20 per graph."""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bluest_amd import spg_device, synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402
from bluest_amd.sap import spg_sap_default_params  # noqa: E402

n, kmax, n_out = (int(a) for a in (sys.argv[1:4] or (20, 5, 8)))
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
            prob["costs"], [prob["costs"]] * n_out, verbose=False)
plan = mos.plan
dev = plan.device
prm = dict(spg_sap_default_params)
w = torch.from_numpy(np.asarray(prob["costs"], dtype=np.float64)).to(dev)
scale = prob["budget"] / w
d = spg_device.DeviceSpg(plan, scale, np.ones(n_out), 32.0 if n_out > 1 else np.inf, prm["scaling_floor"], lmbda_max=prm["lmbda_max"],
                         slots=prm["slots"], check_every=prm["check_every"])
x0 = torch.full((plan.L,), 1.0 / plan.L, dtype=torch.float64, device=dev)
r = d.run(x0, maxit=40)          # captures the graphs
print("warm-up run: it", r["it"], "f", r["f"])
from bluest_amd._lib import check  # noqa: E402
check(plan.lib.bluest_plan_set_gate(plan._h, d.enable.data_ptr(), 1))


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


saved = d.st.clone()
d.st[spg_device.DONE] = 1.0
print("iteration graph, all kernels predicated off: %.1f us per replay" % timed(d.graphs["iteration"].replay, 2000))
d.st.copy_(saved)
print("iteration graph, live:                       %.1f us per replay" % timed(d.graphs["iteration"].replay, 2000))
saved = d.st.clone()
g20 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g20, capture_error_mode="thread_local"):
    for _ in range(20):
        d._iteration()
print("20 iterations per graph, live:               %.1f us per iteration" % (timed(g20.replay, 100) / 20))
d.st[spg_device.DONE] = 1.0
print("20 iterations per graph, predicated off:     %.1f us per iteration" % (timed(g20.replay, 100) / 20))
check(plan.lib.bluest_plan_set_gate(plan._h, None, 0))
