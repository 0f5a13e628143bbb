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
