"""Experiment: the warm MOSAP construction right after the first solve of a process, split into C entry points and the rest."""
import gc
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
from bluest_amd import _lib, synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

n, kmax, n_out = 20, 5, 8
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
torch.zeros(1, device="cuda")
L = _lib.lib()
acc = {}
for name in ("bluest_plan_add_output_cov", "bluest_plan_finalize", "bluest_plan_destroy"):
    fn = getattr(L, name)

    def wrap(*a, _fn=fn, _name=name):
        t0 = time.perf_counter()
        r = _fn(*a)
        acc.setdefault(_name, []).append(time.perf_counter() - t0)
        return r
    setattr(L, name, wrap)
mos = None
for rep in range(3):
    mos = None
    gc.collect()
    gc.disable()
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                prob["costs"], [prob["costs"]] * n_out, verbose=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    gc.enable()
    print("rep %d: construction %.1f ms (+ sync %.1f ms); add_output_cov %s ms; finalize %s ms" % (
        rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, [round(x * 1e3, 1) for x in acc.get("bluest_plan_add_output_cov", [])],
        [round(x * 1e3, 1) for x in acc.get("bluest_plan_finalize", [])]), flush=True)
    mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
