#!/usr/bin/env python3
"""SAP wall-clock (second half of BASELINE.json's metric): covariance -> continuous optimum m*, setup included.
    python tools/sap_wallclock.py [n kmax n_out]
Also prints the PCIe-inclusive rate of the operator when handed numpy arrays."""
import gc
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

n, kmax, n_out = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (20, 5, 8)
solver_params = {kv.split("=")[0]: eval(kv.split("=")[1]) for kv in sys.argv[4:]} or None     # e.g. polish_slots=3
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
out = {"n": n, "kmax": kmax, "n_out": n_out, "K_tot": prob["K_tot"]}
mos = None
for rep in range(3):        # rep0 = cold (first launches, hipGraph captures), rep1, rep2 = warm; nothing is hidden
    mos = None              # release the previous problem
    gc.collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                prob["costs"], [prob["costs"]] * n_out, verbose=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, solver_params=solver_params)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out["rep%d" % rep] = {"setup_s": t1 - t0, "solve_s": t2 - t1, "total_s": t2 - t0, "max_V": max(mos.variances(m)),
                          "nnz": int((m > 1e-9 * m.max()).sum()), "info": {k: (v if isinstance(v, (int, str, bool)) else float(v)) for k, v in mos.solver_info.items() if np.ndim(v) == 0 and not isinstance(v, dict)}}
# PCIe-inclusive operator rate with numpy in / numpy out
mh = prob["m"][0]
mos.variance_GH(mh, nohess=True)
t0 = time.perf_counter()
R = 200
for _ in range(R):
    mos.variance_GH(mh, nohess=True)
out["numpy_in_out_variance_GH_ms"] = (time.perf_counter() - t0) / R * 1e3
md = torch.from_numpy(mh).cuda()
mos.plan.eval(md)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    mos.plan.eval(md)
torch.cuda.synchronize()
out["device_resident_eval_eager_ms"] = (time.perf_counter() - t0) / R * 1e3
print(json.dumps(out, indent=1))
