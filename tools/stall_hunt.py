"""Experiment: the ~65 ms stall that lands in every second warm solve of bench.py's sap_wallclock loop."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP

if os.environ.get("THP_OFF"):
    import ctypes as _ct
    print("prctl(PR_SET_THP_DISABLE) ->", _ct.CDLL(None).prctl(41, 1, 0, 0, 0))
mode = sys.argv[1] if len(sys.argv) > 1 else "asis"
from bluest_amd import spg_device, plan as plan_mod
acc = []


def timed(cls, name, label):
    fn = getattr(cls, name)

    def wrap(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc.append((label, round((time.perf_counter() - t0) * 1e3, 1)))
        return r
    setattr(cls, name, wrap)


_orig_run = spg_device.DeviceSpg.run
first_windows = []


def _run(self, *a, **k):
    t0 = time.perf_counter()
    r = _orig_run(self, *a, **k)
    acc.append(("run", round((time.perf_counter() - t0) * 1e3, 1)))
    if self.L > 4096:
        first_windows.append([round(x * 1e3, 2) for x in self.window_seconds])
        del self.window_seconds[:]
    return r


spg_device.DeviceSpg.run = _run
timed(spg_device.DeviceSpg, "__init__", "spg_init")
timed(plan_mod.Plan, "__init__", "plan_init")
timed(plan_mod.Plan, "eval", "eval")
prob = synth.problem(20, 5, 8)
groups, n_out, kmax = prob["groups"], 8, 5
torch.zeros(1, device="cuda")
sw = None
if os.environ.get("STACKWATCH"):
    import ctypes
    sw = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "stackwatch.so"))
    sw.stackwatch_start(3000)
keepers = []
for rep in range(int(os.environ.get('REPS', '7'))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                prob["costs"], [prob["costs"]] * n_out, verbose=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if mode == "sleep_after_setup":
        time.sleep(0.15)
        t1 = time.perf_counter()
    if sw and rep >= 1:
        sw.stackwatch_arm(1)
    if mode == "twice":
        mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
        torch.cuda.synchronize()
        del first_windows[:]
        t1 = time.perf_counter()
    m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
    if sw and rep >= 1:
        sw.stackwatch_arm(0)
        print("    stack samples:", sw.stackwatch_samples(), flush=True)
        sw.stackwatch_dump(5, 2)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    v = max(mos.variances(m))
    t3 = time.perf_counter()
    if mode == "keep":
        keepers.append(mos)
    mos = None
    if mode == "sleep_between":
        time.sleep(0.15)
    if mode == "asis":
        gc.collect()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("   ", [a for a in acc if a[1] >= 2.0], "n_eval", sum(1 for a in acc if a[0] == "eval"), "eval total", round(sum(a[1] for a in acc if a[0] == "eval"), 1))
    del acc[:]
    print("    windows of the full-problem run (ms):", first_windows)
    del first_windows[:]
    try:
        thr = [l.split()[1] for l in open("/sys/fs/cgroup/cpu.stat") if l.startswith("nr_throttled")][0]
    except Exception:
        thr = "?"
    print("%s rep %d: setup %.1f ms, solve %.1f ms, release %.1f ms  (cgroup nr_throttled %s)" % (mode, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3, thr), flush=True)
