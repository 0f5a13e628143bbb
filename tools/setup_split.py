"""Where the set-up time of a MOSAP goes (second construction, warm): python-side pieces and the C entry points."""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bluest_amd import _lib, synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

n, kmax, n_out = (int(a) for a in (sys.argv[1:4] or (20, 5, 8)))
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]


def mk():
    return MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                 prob["costs"], [prob["costs"]] * n_out, verbose=False)


mk()
torch.cuda.synchronize()
L = _lib.lib()
acc = {}
for name in ("bluest_plan_create", "bluest_plan_add_output_cov", "bluest_plan_finalize", "bluest_plan_destroy"):
    fn = getattr(L, name)

    def wrap(*a, _fn=fn, _name=name):
        t0 = time.perf_counter()
        r = _fn(*a)
        acc[_name] = acc.get(_name, 0.0) + time.perf_counter() - t0
        return r
    setattr(L, name, wrap)
t0 = time.perf_counter()
m = mk()
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("MOSAP construction %.1f ms; inside C entry points: %s; python + rest %.1f ms" % (
    tot * 1e3, {k: round(v * 1e3, 1) for k, v in acc.items()}, (tot - sum(acc.values())) * 1e3))
