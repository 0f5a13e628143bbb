"""Batched throughput (n_cand allocation vectors per launch sequence) and its parts on the chain clock:
    python tools/batch_bench.py [n k n_out]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.plan import Plan, _stream  # noqa: E402

n, k, o = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else bench.HEADLINE
prob = synth.problem(n, k, o)
L = prob["K_tot"]
plan = Plan(n, L, bench.build_outputs(prob), max_candidates=16)
ab = synth.algorithmic_bytes(n, k)
for nc in (1, 4, 16):
    M = torch.from_numpy(10.0 * np.random.RandomState(nc).rand(nc, L)).cuda()
    var = torch.empty((nc, o), dtype=torch.float64, device="cuda")
    grad = torch.empty((nc, plan.grad_len), dtype=torch.float64, device="cuda")
    st = torch.empty((nc, o), dtype=torch.int32, device="cuda")
    vws = torch.empty((nc, o, n), dtype=torch.float64, device="cuda")
    rec = torch.empty((nc, o, plan.reclen), dtype=torch.float64, device="cuda")
    t_all = bench.chain_time(torch, lambda: plan.eval(M, out=(var, grad, st)), R=20)
    t_chunks = bench.chain_time(torch, lambda: plan.lib.bluest_plan_phi_chunks(plan._h, M.data_ptr(), nc, L, _stream()), R=20)
    t_nograd = bench.chain_time(torch, lambda: plan.eval(M, want_grad=False, out=(var, None, st)), R=20)
    plan.solve(plan.phi(M, out=rec), out=(var, vws, st))
    t_grad = bench.chain_time(torch, lambda: plan.grad(vws, st, out=grad), R=20)
    moved = plan.phi_bytes + plan.grad_bytes + nc * (L * 8 + plan.grad_len * 8 + 2 * (plan.phi_bytes // (256 * 12)) * 16)
    print("n=%d k<=%d n_out=%d  n_cand=%2d: sequence %.2f us = %.3f M assemblies/s | Phi chunks %.2f, solve %.2f, gradient %.2f us | moved %.1f MB -> %.2f of 8 TB/s"
          % (n, k, o, nc, t_all * 1e6, nc * o / t_all / 1e6, t_chunks * 1e6, (t_nograd - t_chunks) * 1e6, t_grad * 1e6, moved / 1e6, moved / t_all / 8e12), flush=True)
