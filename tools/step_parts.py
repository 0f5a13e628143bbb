"""chain-clock durations of the parts of one evaluation step (hipGraph of 50 launches, HIP events): Phi chunks, the fused
solve + gradient from the chunk partials (the hot path), and the same kernel fed from a ready Phi record"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
sys.argv = sys.argv[:1] + sys.argv[1:]
import bench  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.plan import Plan, _stream  # noqa: E402

n, kmax, n_out = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else bench.HEADLINE
prob = synth.problem(n, kmax, n_out)
plan = Plan(n, prob["K_tot"], bench.build_outputs(prob), max_candidates=2)
dev = plan.device
L = prob["K_tot"]
m = torch.from_numpy(prob["m"][0]).to(dev)
var = torch.empty((1, n_out), dtype=torch.float64, device=dev)
grad = torch.empty((1, plan.grad_len), dtype=torch.float64, device=dev)
status = torch.empty((1, n_out), dtype=torch.int32, device=dev)
rec = plan.phi(m)
lib, h = plan.lib, plan._h
t_chunks = bench.chain_time(torch, lambda: lib.bluest_plan_phi_chunks(h, m.data_ptr(), 1, L, _stream()))
t_step = bench.chain_time(torch, lambda: plan.eval(m, out=(var, grad, status)))
t_rec = bench.chain_time(torch, lambda: plan.solve_grad(rec, out=(var, grad, status)))
t_phi = bench.chain_time(torch, lambda: plan.phi(m, out=rec))
print("chunks %.2f us  step %.2f us  solve_grad(partials) %.2f us  solve_grad(record) %.2f us  phi(chunks+fold_to_record) %.2f us"
      % (t_chunks * 1e6, t_step * 1e6, (t_step - t_chunks) * 1e6, t_rec * 1e6, t_phi * 1e6))
