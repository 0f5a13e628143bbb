"""
Experiment: per-phase timestamps of k_solve_grad workgroups (first / middle / last of the grid).  Needs a library built with
    BLUEST_EXTRA_HIPCC_FLAGS=-DBLUEST_PHASE_TIMING python -m bluest_amd.build --force
(never the shipped build).  Prints phase durations in microseconds (s_memtime = shader cycles, converted at 2.4 GHz; every
stamp costs a few tens of cycles itself, far less than the s_memrealtime stamps used before).
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
L.bluest_debug_phase_times.argtypes = [ctypes.c_void_p]
acc = []
for rep in range(30):
    plan.eval(m)
    torch.cuda.synchronize()
    t = np.zeros((3, 12), dtype=np.int64)
    assert L.bluest_debug_phase_times(t.ctypes.data) == 0
    if rep >= 5:
        acc.append(t.astype(np.float64))
a = np.median(np.array(acc), axis=0) / 2400.0   # us (s_memtime counts shader cycles; 2.4 GHz)
names = ["zero+desc", "fold", "solve", "grad tile"]
for b, who in enumerate(["first wg", "middle wg", "last wg"]):
    print(who, "start %+.2f us (rel. first wg)" % (a[b, 0] - a[0, 0]), {names[i]: round(a[b, i + 1] - a[b, i], 2) for i in range(4)},
          "total %.2f" % (a[b, 4] - a[b, 0]),
          "| pre-solve %.2f, masks %.2f, load %.2f |" % (a[b, 8] - a[b, 2], a[b, 9] - a[b, 8], a[b, 5] - a[b, 9]),
          "| solve: masks+load %.2f, factorise %.2f, back-substitution %.2f, publish %.2f" % (a[b, 5] - a[b, 2], a[b, 6] - a[b, 5], a[b, 7] - a[b, 6], a[b, 3] - a[b, 7]))
