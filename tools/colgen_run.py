"""second-order finish alone on a synthetic problem: python tools/colgen_run.py n kmax n_out [key=value ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.colgen import colgen_solve  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

n, kmax, n_out = (int(a) for a in sys.argv[1:4])
prm = {}
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    prm[k] = eval(v)
verbose = prm.pop("verbose", False)
ncaps = prm.pop("caps", 0)
prob = synth.problem(n, kmax, n_out)
groups = prob["groups"]
mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
            prob["costs"], [prob["costs"]] * n_out, verbose=False)
caps = None
if ncaps:        # the ncaps most sampled models capped at half of what the free optimum gives them
    x_free, _ = colgen_solve(mos.plan, prob["costs"], np.ones(n_out), prob["budget"], prm=prm)
    m_free = prob["budget"] / prob["costs"] * x_free
    usage = np.array([float(mos.ES[i] @ m_free) for i in range(n)])
    models = np.sort(np.argsort(-usage)[:ncaps])
    caps = {"models": models, "rows": np.stack([mos.ES[i] for i in models]), "rhs": np.array([max(1.0, np.floor(0.5 * usage[i])) for i in models])}
    print("caps:", models, caps["rhs"])
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, info = colgen_solve(mos.plan, prob["costs"], np.ones(n_out), prob["budget"], prm=prm, log=print if (verbose and rep == 0) else None, caps=caps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if x is None:
        print("failed:", info)
        break
    m = prob["budget"] / prob["costs"] * x
    print("rep %d: %.4f s  max V %.12e  nnz %d  %s" % (rep, dt, max(mos.variances(m)), int((x > 0).sum()),
                                                     {k: (float("%.4g" % v) if isinstance(v, float) else v) for k, v in info.items() if k not in ("mu", "certificate", "cap_usage")}))
if caps is not None:         # the same caps through the free solver under shifted costs (bluest_amd.capped.cost_shift_capped)
    from bluest_amd.capped import cost_shift_capped
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        from bluest_amd.host import host_section
        with host_section():        # (as MOSAP.solve does: no BLAS thread pool, no cyclic collector while the kernels are driven)
            m, info = cost_shift_capped(mos.plan, prob["costs"], np.ones(n_out), prob["budget"], caps["rows"], caps["rhs"], prm=dict(prm),
                                        log=print if (verbose and rep == 0) else None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if m is None:
            print("cost shift failed:", info)
            break
        print("cost shift rep %d: %.4f s  max V %.12e  nnz %d  %s" % (rep, dt, max(mos.variances(m)), int((m > 0).sum()),
              {k: (float("%.4g" % v) if isinstance(v, float) else v) for k, v in info.items() if k not in ("mu", "certificate")}))

