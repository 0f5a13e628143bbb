"""
Worker of tests/test_gpu_dist.py (launched with torch.distributed.run, one process per rank, ALL ranks on cuda:0 with the gloo
backend -- RCCL refuses two ranks per device; on a multi-GPU node the same code runs with one rank per GPU over RCCL).
Group-sharded HIP plans: evaluation (Phi records all-reduced, redundant solves, per-shard gradients) and the sharded SPG solve;
rank 0 writes what the test compares with the single-process results.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bluest_amd import synth                      # noqa: E402
from bluest_amd.dist import ShardedPlan, sharded_spg   # noqa: E402
from bluest_amd.plan import Plan                  # noqa: E402


def main(out_path):
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    res = {"world": world}
    # the one-shot peer-write exchange (C-ABI Part 5) against the group's all_reduce, incl. back-to-back calls of different length
    from bluest_amd.dist import PeerExchange
    ex = PeerExchange.create(2000, dev)
    res["peer_exchange_available"] = ex is not None
    if ex is not None:
        rng = np.random.RandomState(5 + rank)
        worst, same = 0.0, True
        for n_ in (1, 63, 676, 2000, 7, 2000, 1999):
            a = torch.from_numpy(rng.randn(n_)).to(dev)
            b = a.clone()
            ex.all_reduce(a)
            dist.all_reduce(b)
            torch.cuda.synchronize()
            worst = max(worst, float((a - b).abs().max()))
            gathered = [torch.empty(n_, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(gathered, a.cpu())
            same = same and all(torch.equal(gathered[0], q) for q in gathered)
        res["peer_exchange_err"], res["peer_exchange_identical_on_all_ranks"], res["peer_exchange_timed_out"] = worst, same, ex.timed_out()
    # (the third problem is long enough, K_tot = 6884 > 4096, for the solver's working set: replicated restricted plans,
    # collective pricing)
    cases = (("n12_k4_o1", (12, 4, 1)), ("n10_k3_o3", (10, 3, 3)), ("n16_k5_o2", (16, 5, 2)))
    if os.environ.get("BLUEST_TEST_BIG"):          # BASELINE.json configs[4] across the process boundary (n = 25, K_tot = 245505)
        cases = (("n25_k6_o1", (25, 6, 1)),)
    for tag, (n, kmax, n_out) in cases:
        prob = synth.problem(n, kmax, n_out)
        sizes = [len(g) for g in prob["groups"]]
        outs = [{"K": kmax, "sizes": sizes, "groups": prob["groups"], "C": prob["C"][o], "mapping": None} for o in range(n_out)]
        sp = ShardedPlan(n, sizes, outs, device=dev)
        full = Plan(n, prob["K_tot"], outs, device=dev)
        m = torch.from_numpy(prob["m"][0]).to(dev)
        var, grad_local, status = sp.eval(m)
        v_full, g_full, st_full = full.eval(m)
        coef = torch.from_numpy(np.linspace(0.2, 1.0, n_out).reshape(1, -1)).to(dev)
        g_glob = sp.combine_grad(grad_local, coef)
        g_want = full.combine_grad(g_full, coef)
        res[tag + "_eval_err"] = float((var / v_full - 1).abs().max())
        res[tag + "_grad_err"] = float((g_glob - g_want).abs().max() / g_want.abs().max())
        res[tag + "_status_equal"] = bool(torch.equal(status, st_full))
        res[tag + "_shard"] = [sp.lo, sp.hi]
        res[tag + "_exchange"] = sp.exchange_name
        big = n >= 25
        # host-driven driver (collective callbacks) and the device-resident collective loop (the default on GPUs)
        m_host, info_h = (np.zeros(prob["K_tot"]), None) if big else sharded_spg(sp, prob["costs"], budget=prob["budget"], params={"smoothing_p": 512.0, "device_loop": False, "method": "spg"})
        if not big:
            vh, _, _ = full.eval(torch.from_numpy(m_host).to(dev), want_grad=False)
            res[tag + "_F_sharded_host_loop"] = float(vh.max())
        m_sh, info = sharded_spg(sp, prob["costs"], budget=prob["budget"])
        all_m = [torch.empty(prob["K_tot"], dtype=torch.float64) for _ in range(world)]
        dist.all_gather(all_m, torch.from_numpy(np.ascontiguousarray(m_sh)))
        res[tag + "_ranks_agree"] = bool(all(torch.equal(all_m[0], q) for q in all_m))
        vs, _, st = full.eval(torch.from_numpy(m_sh).to(dev), want_grad=False)
        res[tag + "_F_sharded"] = float(vs.max())
        res[tag + "_cost_ratio"] = float(m_sh @ prob["costs"] / prob["budget"])
        res[tag + "_it"] = int(info["it"])
        res[tag + "_support"] = int((m_sh > 0).sum())
        res[tag + "_method"] = info.get("method", "spg")
        res[tag + "_gap"] = float(info.get("certified_gap", np.nan))
        if not big:                                # the first-order loop on replicated vectors stays available
            m_fo, info_fo = sharded_spg(sp, prob["costs"], budget=prob["budget"], params={"method": "spg"})
            vf, _, _ = full.eval(torch.from_numpy(m_fo).to(dev), want_grad=False)
            res[tag + "_F_sharded_first_order"] = float(vf.max())
        else:                                      # single-GPU answer of the same problem, computed by rank 0's process too
            from bluest_amd.colgen import colgen_solve
            x1, i1 = colgen_solve(full, prob["costs"], np.ones(n_out), prob["budget"])
            v1, _, _ = full.eval(torch.from_numpy(prob["budget"] / prob["costs"] * x1).to(dev), want_grad=False)
            res[tag + "_F_single"] = float(v1.max())
    if rank == 0:
        json.dump(res, open(out_path, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
