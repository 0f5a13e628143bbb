"""
GPU test of the N > 1 path on REAL HIP plans: two ranks (two processes, both on the one GPU of the test box, gloo backend) run
the group-sharded evaluation and the sharded SPG solve of bluest_amd/dist.py; compared with the single-process plan / solver.
(The CPU twin of the wiring is tests/test_dist_gloo.py.)
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from bluest_amd import synth
from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_sharded_hip_plans_two_ranks(tmp_path):
    out = str(tmp_path / "sharded.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "tests", "sharded_worker.py"), out]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    res = json.load(open(out))
    assert res["world"] == 2
    # hipIpc-shared fine-grained mailboxes work between two processes of this box: the custom exchange must be in use and exact
    assert res["peer_exchange_available"], proc.stderr[-2000:]
    assert res["peer_exchange_err"] < 1e-13 and res["peer_exchange_identical_on_all_ranks"] and not res["peer_exchange_timed_out"]
    assert res["n12_k4_o1_exchange"].startswith("peer-write")
    from bluest_amd.mosap import MOSAP
    for tag, (n, kmax, n_out) in (("n12_k4_o1", (12, 4, 1)), ("n10_k3_o3", (10, 3, 3)), ("n16_k5_o2", (16, 5, 2))):
        assert res[tag + "_eval_err"] < 1e-12 and res[tag + "_grad_err"] < 1e-12 and res[tag + "_status_equal"]
        assert res[tag + "_ranks_agree"] and abs(res[tag + "_cost_ratio"] - 1) < 1e-9
        lo, hi = res[tag + "_shard"]
        assert lo == 0 and 0 < hi < synth.n_groups(n, kmax)
        # the sharded solve (second-order finish: sharded multiplicative phase, collective column generation, redundant masters)
        # reaches the single-GPU optimum with a certified gap; the first-order loops (device-resident on replicated vectors,
        # host-driven with collective callbacks) stay available and get close
        prob = synth.problem(n, kmax, n_out)
        groups = prob["groups"]
        mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)],
                    prob["costs"], [prob["costs"]] * n_out, verbose=False)
        m1 = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
        F1 = max(mos.variances(m1))
        assert res[tag + "_method"] == "newton" and res[tag + "_gap"] <= 1e-6, (tag, res[tag + "_method"], res[tag + "_gap"])
        assert abs(res[tag + "_F_sharded"] / F1 - 1) < 1e-6, (tag, res[tag + "_F_sharded"], F1, res[tag + "_it"])
        assert res[tag + "_support"] <= 4 * n
        assert res[tag + "_F_sharded_first_order"] <= F1 * (1 + 2e-3), (tag, res[tag + "_F_sharded_first_order"], F1)
        assert res[tag + "_F_sharded_host_loop"] <= F1 * (1 + 5e-3), (tag, res[tag + "_F_sharded_host_loop"], F1)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_solve_n25_across_processes(tmp_path, world):
    """BASELINE.json configs[4] (n = 25, k_max = 6, K_tot = 245505) with the group set cut over 2 and 4 processes (all on the one
    GPU of the test box, gloo bootstrap, peer-write record exchange): the sharded second-order finish ends at the single-GPU
    optimum with a certified gap, all ranks return identical bits, and nothing of length K_tot is exchanged per step"""
    out = str(tmp_path / "sharded25.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BLUEST_TEST_BIG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29540 + world), os.path.join(ROOT, "tests", "sharded_worker.py"), out]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    res = json.load(open(out))
    tag = "n25_k6_o1"
    assert res["world"] == world and res["peer_exchange_available"] and res[tag + "_exchange"].startswith("peer-write")
    assert res[tag + "_eval_err"] < 1e-12 and res[tag + "_grad_err"] < 1e-12 and res[tag + "_status_equal"]
    assert res[tag + "_ranks_agree"] and abs(res[tag + "_cost_ratio"] - 1) < 1e-9
    assert res[tag + "_method"] == "newton" and res[tag + "_gap"] <= 1e-6
    assert abs(res[tag + "_F_sharded"] / res[tag + "_F_single"] - 1) < 1e-6, (res[tag + "_F_sharded"], res[tag + "_F_single"])


def test_rccl_world_size_one(tmp_path):
    """RCCL itself on this box: init_process_group("nccl") with ONE rank, a sharded evaluation through the group's all_reduce
    (exchange="collective") and a plain dist.all_reduce -- the library initialises and one RCCL kernel has run (two ranks per
    device are refused by RCCL, and the pool has single-GPU boxes; the N-GPU runs are the driver's)"""
    code = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from bluest_amd import synth
from bluest_amd.dist import ShardedPlan, sharded_spg
from bluest_amd.plan import Plan
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(5, dtype=torch.float64, device=dev)
dist.all_reduce(t)
prob = synth.problem(10, 3, 2)
sizes = [len(g) for g in prob["groups"]]
outs = [{"K": 3, "sizes": sizes, "groups": prob["groups"], "C": prob["C"][o], "mapping": None} for o in range(2)]
sp = ShardedPlan(10, sizes, outs, device=dev, exchange="collective")
rec = sp.plan.phi(prob["m"][0])
dist.all_reduce(rec)                                   # the collective the N-GPU path issues per evaluation (RCCL kernel)
var, g, st = sp.eval(prob["m"][0])
full = Plan(10, prob["K_tot"], outs, device=dev)
v2, g2, _ = full.eval(prob["m"][0])
m, info = sharded_spg(sp, prob["costs"], budget=prob["budget"])
torch.cuda.synchronize()
print(json.dumps({"sum": float(t.sum()), "err": float((var / v2 - 1).abs().max()), "backend": dist.get_backend(),
                  "method": info.get("method"), "gap": float(info.get("certified_gap", 1.0))}))
dist.destroy_process_group()
''' % ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-3000:]
    res = json.loads(proc.stdout.strip().splitlines()[-1])
    assert res["backend"] == "nccl" and res["sum"] == 5.0 and res["err"] < 1e-12
    assert res["method"] == "newton" and res["gap"] <= 1e-6
