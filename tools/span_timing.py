"""
Experiment: timeline of a chain of evaluations -- begin/end of the Phi chunk kernel and of the fused solve+gradient
kernel for 12 back-to-back evaluations (100 MHz wall clock, sampled workgroups).  Needs the BLUEST_PHASE_TIMING build
(see tools/phase_timing.py).
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # noqa: E402
from bluest_amd import _lib, synth             # noqa: E402
from bluest_amd.plan import Plan               # noqa: E402

n, kmax, n_out = (int(a) for a in (sys.argv[1:4] or (20, 5, 8)))
dev = torch.device("cuda", 0)
prob = synth.problem(n, kmax, n_out)
plan = Plan(n, prob["K_tot"], bench.build_outputs(prob), max_candidates=1, device=dev)
m = torch.from_numpy(prob["m"][0]).to(dev)
L = _lib.lib()
L.bluest_debug_span_read.argtypes = [ctypes.c_void_p]
var = torch.empty((1, plan.n_out), dtype=torch.float64, device=dev)
grad = torch.empty((1, plan.grad_len), dtype=torch.float64, device=dev)
status = torch.empty((1, plan.n_out), dtype=torch.int32, device=dev)
g = torch.cuda.CUDAGraph()
for _ in range(3):
    plan.eval(m, out=(var, grad, status))
torch.cuda.synchronize()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    for _ in range(12):
        plan.eval(m, out=(var, grad, status))
for rep in range(3):
    g.replay()
torch.cuda.synchronize()
assert L.bluest_debug_span_reset() == 0
g.replay()
torch.cuda.synchronize()
t = np.zeros((16, 2, 2), dtype=np.uint64)
assert L.bluest_debug_span_read(t.ctypes.data) == 0
t = t[:12].astype(np.float64) * 0.01
t0 = t[0, 0, 0]
print("step | chunks begin  end (dur) | gap | solve+grad begin  end (dur) | gap to next step | step total")
for i in range(12):
    cb, ce, fb, fe = t[i, 0, 0] - t0, t[i, 0, 1] - t0, t[i, 1, 0] - t0, t[i, 1, 1] - t0
    nxt = t[i + 1, 0, 0] - t0 if i + 1 < 12 else float("nan")
    print("%4d | %8.2f %8.2f (%5.2f) | %5.2f | %8.2f %8.2f (%5.2f) | %5.2f | %6.2f" % (i, cb, ce, ce - cb, fb - ce, fb, fe, fe - fb, nxt - fe, nxt - cb))
