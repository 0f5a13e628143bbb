"""Kernel time of the scaled simplex projection step (as the device SPG calls it) for a few vector lengths: 50 projections
per hipGraph, inputs drifting slowly so the warm start is realistic.  BLUEST_PROJ_MULTI_LAUNCH=1 selects the multi-launch path."""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bluest_amd.plan import simplex_project  # noqa: E402

dev = torch.device("cuda")
rng = np.random.RandomState(0)
for L in [int(a) for a in sys.argv[1:]] or [21699, 245505]:
    x = torch.from_numpy(rng.dirichlet(np.full(L, 0.05))).to(dev)
    gs = [torch.from_numpy(rng.randn(L) * (1 + 0.01 * k)).to(dev) for k in range(5)]
    for _ in range(3):
        simplex_project(x, gs[0], 1e-3, floor=1e-8)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for k in range(50):
            out = simplex_project(x, gs[k % 5], 1e-3 * (1 + 0.1 * (k % 3)), floor=1e-8)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R):
        g.replay()
    torch.cuda.synchronize()
    from bluest_amd.plan import projection_workspace
    ws = projection_workspace(L, x.device)
    off = 2 * L + 4 * ((L + 1023) // 1024)
    t = ws[off:off + 16].cpu().numpy()
    print("L = %7d: %.2f us per projection  (sum p = %.15f)  passes per search %.2f" % (
        L, (time.perf_counter() - t0) / (R * 50) * 1e6, float(out[0].sum()), t[9] / max(t[10], 1)))
