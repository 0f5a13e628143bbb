"""Per-run log of the device SPG inside SAP/MOSAP.solve: iterations, evaluations, seconds, objective for every
(continuation stage, restart).   python tools/solve_trace.py n kmax n_out [key=value ...solver params]"""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: E402
from bluest_amd import spg_device, synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

n, kmax, n_out = (int(a) for a in sys.argv[1:4])
params = {}
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    params[k] = eval(v)
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
            prob["costs"], [prob["costs"]] * n_out, verbose=False)
orig = spg_device.DeviceSpg.run


def logged(self, x0, **kw):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = orig(self, x0, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L = self.L
    if L > 4096:   # multi-CU projection path keeps search statistics in its workspace (csrc: ProjWs)
        off = 2 * L + 4 * ((L + 1023) // 1024)
        t = self.pws[off:off + 16].cpu().numpy()
        print("  threshold searches %d, passes per search %.2f" % (t[10], t[9] / max(t[10], 1.0)))
        self.pws[off + 9:off + 11] = 0.0
    print("  run p=%-6g it %5d evals %6d  %.4f s  (%.1f us/it)  f %.9g gpmax %.2e info %d stalled %s" % (
        self.p, r["it"], r["count"], dt, 1e6 * dt / max(r["it"], 1), r["f"], r["gpmax"], r["solver_info"], r["stalled"]))
    return r


spg_device.DeviceSpg.run = logged
orig_capture = spg_device.DeviceSpg._capture


def timed_capture(self, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = orig_capture(self, fn)
    torch.cuda.synchronize()
    print("    capture %-20s L=%d T=%d: %.1f ms" % (fn.__name__, self.L, self.T, 1e3 * (time.perf_counter() - t0)))
    return g


spg_device.DeviceSpg._capture = timed_capture
import gc
run_total = [0.0]
_logged = spg_device.DeviceSpg.run


def summed(self, x0, **kw):
    t0 = time.perf_counter()
    r = _logged(self, x0, **kw)
    run_total[0] += time.perf_counter() - t0
    return r


spg_device.DeviceSpg.run = summed
for rep in range(3):
    gc.collect()
    gc.disable()
    run_total[0] = 0.0
    t0 = time.perf_counter()
    m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, solver_params=params or None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    print("solve %.4f s (inside DeviceSpg.run incl. the trace's own syncs: %.4f s)  max V %.9g  nnz %d" % (
        dt, run_total[0], max(mos.variances(m)), int((m > 0).sum())), mos.solver_info)
