"""Certified gap, warm solve time and the polish passes of the second-order finish on the BASELINE configurations and on the larger ones.
    python tools/gap_table.py [key=value ...]        (solver parameters as tools/colgen_run.py takes them)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.colgen import colgen_solve  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

prm = {kv.split("=")[0]: eval(kv.split("=", 1)[1]) for kv in sys.argv[1:]}
print("# parameters: %s" % (prm or "defaults"))
print("# n kmax n_out | warm solve (best of 3) | certified gap | rounds + final rounds | polish passes (support, status, Newton iterations, KKT measure)")
for n, k, o in ((12, 12, 1), (20, 5, 1), (20, 5, 8), (25, 6, 1), (16, 4, 3), (30, 3, 4), (48, 2, 1)):
    prob = synth.problem(n, k, o)
    g = prob["groups"]
    mos = MOSAP(prob["C"], k, [k] * o, [a.copy() for a in g], [[a.copy() for a in g] for _ in range(o)], prob["costs"], [prob["costs"]] * o, verbose=False)
    best, info = np.inf, None
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, info = colgen_solve(mos.plan, prob["costs"], np.ones(o), prob["budget"], prm=dict(prm))
        torch.cuda.synchronize()
        if rep:
            best = min(best, time.perf_counter() - t0)
    if x is None:
        print("%2d %2d %d | failed: %s" % (n, k, o, info))
        continue
    print("%2d %2d %d | %6.2f ms | gap %+.2e | %2d + %d | %s" % (n, k, o, best * 1e3, info["gap"], info["rounds"], info.get("final_rounds", 0),
          [(q["support"], q["status"], q["newton_it"], float("%.1e" % q["kkt"])) for q in info["polish_passes"]]), flush=True)
