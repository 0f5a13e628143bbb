"""Average number of Newton passes (= mailbox exchanges) per multi-workgroup projection during a solve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
from bluest_amd.plan import _WORKSPACES
for n, k, o in ((20, 5, 8), (25, 6, 1)):
    prob = synth.problem(n, k, o)
    g = prob["groups"]
    mos = MOSAP(prob["C"], k, [k] * o, [a.copy() for a in g], [[a.copy() for a in g] for _ in range(o)], prob["costs"], [prob["costs"]] * o, verbose=False)
    m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
    for (L, dev), ws in _WORKSPACES.items():
        if L != mos.L:
            continue
        nb = (L + 1023) // 1024
        off = 2 * L + 4 * max(nb, 64)
        t = ws[off:off + 16].cpu().numpy()
        print("K_tot %d: %d projections, %.2f passes on average" % (L, int(t[10]), t[9] / max(t[10], 1)))
