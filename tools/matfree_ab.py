"""Matrix-free against stored-inverse evaluation, same box, same inputs: step time on the chain clock (hipGraph of 50 dependent
steps, HIP events) and the parts of the matrix-free step.   python tools/matfree_ab.py [n k n_out ...triples]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd._lib import check  # noqa: E402
from bluest_amd.plan import Plan, _stream  # noqa: E402

args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(25, 6, 1), (20, 5, 8), (20, 5, 1), (16, 4, 3)]
for n, k, o in shapes:
    prob = synth.problem(n, k, o)
    L = prob["K_tot"]
    m = torch.from_numpy(prob["m"][0]).cuda()
    row = {}
    for mode in ("1", "2", "0"):
        os.environ["BLUEST_MATFREE"] = mode
        plan = Plan(n, L, bench.build_outputs(prob))
        var = torch.empty((1, o), dtype=torch.float64, device="cuda")
        grad = torch.empty((1, plan.grad_len), dtype=torch.float64, device="cuda")
        st = torch.empty((1, o), dtype=torch.int32, device="cuda")
        rec = torch.empty((1, o, plan.reclen), dtype=torch.float64, device="cuda")
        t_step = bench.chain_time(torch, lambda: plan.eval(m, out=(var, grad, st)))
        t_phi = bench.chain_time(torch, lambda: plan.phi(m, out=rec))
        plan.phi(m, out=rec)
        t_sg = bench.chain_time(torch, lambda: plan.solve_grad(rec, out=(var, grad, st)))
        row[mode] = (plan.matfree, t_step * 1e6, t_phi * 1e6, t_sg * 1e6, plan.matfree_bytes if plan.matfree else plan.phi_bytes + plan.grad_bytes)
        del plan
    a, h, b = row["1"], row["2"], row["0"]
    print("n=%d k<=%d n_out=%d K_tot=%d | matrix-free(%s): step %.2f us (Phi -> record %.2f, solve + gradient from the record %.2f), %.1f MB moved | "
          "stored Phi + matrix-free gradient: step %.2f us | stored: step %.2f us (Phi -> record %.2f, solve + gradient %.2f), %.1f MB streamed | "
          "stored / matrix-free %.2f, stored / hybrid %.2f"
          % (n, k, o, L, a[0], a[1], a[2], a[3], a[4] / 1e6, h[1], b[1], b[2], b[3], b[4] / 1e6, b[1] / a[1], b[1] / h[1]), flush=True)
