import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bluest_amd.plan import simplex_project, projection_workspace
rng = np.random.RandomState(0)
dev = torch.device("cuda")
for L in (4097, 21699):
    for scale in (1.0, 1e-6, 1e30):
        v = scale * rng.randn(L)
        x = torch.from_numpy(v).to(dev)
        p, d, stats = simplex_project(x)
        ws = projection_workspace(L, dev)
        nb = (L + 1023) // 1024
        off = 2 * L + 4 * nb
        t = ws[off:off + 16].cpu().numpy()
        sy = ws[off + 16: off + 16 + 40].cpu().numpy().view(np.uint32)
        print(L, scale, "sum p", float(p.sum()), "stats", stats.cpu().numpy(), "t", t[:11], "ctr", sy[:6], sy[72:74])
