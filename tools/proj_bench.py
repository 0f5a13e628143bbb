"""Micro-benchmark of the simplex projection alone: single-workgroup kernel (work=None) vs the multi-workgroup mailbox kernel,
at SPG-like inputs (x on the simplex, a gradient step).  Usage: python tools/proj_bench.py [L ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import _lib
from bluest_amd.plan import projection_workspace, check, _stream

lib = _lib.lib()
for L in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 21699, 24576]:
    rng = np.random.default_rng(L)
    xh = rng.random(L) ** 8
    xh /= xh.sum()
    x = torch.from_numpy(xh).cuda()
    g = torch.from_numpy(rng.standard_normal(L) * 1e-3).cuda()
    p, d, stats = torch.empty_like(x), torch.empty_like(x), torch.empty(4, dtype=torch.float64, device="cuda")
    res = {}
    for name, work in (("single", None), ("multi", projection_workspace(L, x.device))):
        if name == "single" and L > 512 * 48:
            continue

        def call():
            check(lib.bluest_simplex_project(x.data_ptr(), g.data_ptr(), 0.5, 1.0, 1e-9, L, p.data_ptr(), d.data_ptr(), stats.data_ptr(),
                                             None if work is None else work.data_ptr(), _stream()))
        for _ in range(20):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500):
            call()
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) * 2.0, p.clone())
    line = "L %6d: " % L + "  ".join("%s %.2f us" % (k, v[0]) for k, v in res.items())
    if len(res) == 2:
        line += "   max |p_single - p_multi| = %.2e" % float((res["single"][1] - res["multi"][1]).abs().max())
    print(line, flush=True)
