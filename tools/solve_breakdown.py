"""Experiment: where the wall-clock of MOSAP.solve(solver="spg") goes (graph captures, device loop, restricted plans, host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import spg_device, synth, mosap as mosap_mod, sap as sap_mod
from bluest_amd.mosap import MOSAP

n, kmax, n_out = (int(a) for a in (sys.argv[1:4] or (20, 5, 8)))
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
acc = {}


def timed(cls, name, label):
    fn = getattr(cls, name)

    def wrap(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc.setdefault(label, []).append(time.perf_counter() - t0)
        return r
    setattr(cls, name, wrap)


names = []
_orig_capture = spg_device.DeviceSpg._capture


def _cap(self, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = _orig_capture(self, fn)
    torch.cuda.synchronize()
    names.append((fn.__name__, self.T, self.L, round((time.perf_counter() - t0) * 1e3, 2)))
    acc.setdefault("graph capture", []).append(time.perf_counter() - t0)
    return g


spg_device.DeviceSpg._capture = _cap
timed(spg_device.DeviceSpg, "run", "DeviceSpg.run (incl. captures)")
timed(spg_device.DeviceSpg, "__init__", "DeviceSpg.__init__")
timed(MOSAP, "_restricted_plan", "restricted plan")
for rep in range(3):
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                prob["costs"], [prob["costs"]] * n_out, verbose=False)
    acc.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print("rep %d: solve %.1f ms, it %d; " % (rep, tot * 1e3, mos.solver_info["it"]) +
          "; ".join("%s: %d calls %.1f ms" % (k, len(v), sum(v) * 1e3) for k, v in acc.items()), flush=True)
    runs = acc.get("DeviceSpg.run (incl. captures)", [])
    print("   runs (ms):", [round(x * 1e3, 1) for x in runs])
    print("   captured:", names)
    del names[:]
    t0 = time.perf_counter()
    mos = None
    import gc
    gc.collect()
    torch.cuda.synchronize()
    print("   release %.1f ms" % ((time.perf_counter() - t0) * 1e3))
