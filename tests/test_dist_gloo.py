"""
CPU test of the N>1 path (world_size 2, gloo): group-set sharding + ONE all-reduce(SUM) of the Phi records +
redundant solve + per-shard gradient, as bluest_amd/dist.py wires it.  The HIP plan cannot run without a GPU, so
the numeric back-end is a CPU stand-in with the same phi/solve/grad contract built on the oracle; what is under
test is the partition (every group owned exactly once, ragged mappings included), the record semantics (sums and
indicators combine by addition) and the collective wiring.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bluest_amd import synth
from bluest_amd.dist import ShardedPlan, shard_bounds, shard_output


class CpuStandInPlan(object):
    """same contract as bluest_amd.plan.Plan (phi -> record, solve, grad, combine_grad) on CPU tensors"""

    def __init__(self, n_models, L_global, outputs, max_candidates=1, device=None):
        from oracle import oracle as orc
        self.orc = orc
        self.N, self.L, self.n_out = n_models, L_global, len(outputs)
        self.outs = []
        self.grad_off, off = [], 0
        for o in outputs:
            K = o["K"]
            groups = [np.asarray(g, dtype=np.int64).reshape(-1, k + 1) for k, g in enumerate(o["groups"])]
            invcovs = [np.concatenate([np.linalg.pinv(o["C"][np.ix_(g, g)]).ravel() for g in gk]) if len(gk) else np.zeros(0)
                       for gk in groups]
            self.outs.append((K, groups, invcovs, np.asarray(o["mapping"], dtype=np.int64)))
            self.grad_off.append(off)
            off += len(o["mapping"])
        self.grad_len = off
        self.reclen = n_models * n_models + 2 * n_models + 1

    def phi(self, m, out=None):
        m = np.asarray(m, dtype=np.float64).reshape(1, -1)
        N = self.N
        rec = np.zeros((1, self.n_out, self.reclen))
        for o, (K, groups, invcovs, mapping) in enumerate(self.outs):
            ml = m[0, mapping]
            PHI, off = np.zeros(N * N), 0
            touched1, touched2 = np.zeros(N), np.zeros(N)
            for k in range(1, K + 1):
                gk = groups[k - 1]
                mk = ml[off:off + len(gk)]
                if len(gk):
                    PHI += self.orc.objectiveK(N, k, len(gk), mk, gk, invcovs[k - 1])
                    touched1[np.unique(gk[np.abs(mk) > 1e-6])] = 1.0
                    touched2[np.unique(gk[mk != 0])] = 1.0
                off += len(gk)
            rec[0, o, :N * N] = PHI
            rec[0, o, N * N:N * N + N] = touched1
            rec[0, o, N * N + N:N * N + 2 * N] = touched2
            rec[0, o, -1] = 1.0 if np.abs(ml).max() >= 0.05 else 0.0
        return torch.from_numpy(rec)

    def solve(self, rec, delta=0.0):
        rec = rec.numpy()
        N = self.N
        var = np.zeros((1, self.n_out)); v = np.zeros((1, self.n_out, N)); status = np.zeros((1, self.n_out), dtype=np.int32)
        for o in range(self.n_out):
            PHI = rec[0, o, :N * N].reshape(N, N) + delta * np.eye(N)
            idx = np.nonzero(rec[0, o, N * N:N * N + N] > 0)[0]
            var[0, o] = np.linalg.solve(PHI[np.ix_(idx, idx)], np.eye(len(idx), 1).ravel())[0]
            v[0, o] = np.linalg.pinv(PHI)[0]
        return torch.from_numpy(var), torch.from_numpy(v), torch.from_numpy(status)

    def grad(self, v, status, out=None):
        g = np.zeros((1, self.grad_len))
        for o, (K, groups, invcovs, mapping) in enumerate(self.outs):
            off = self.grad_off[o]
            for k in range(1, K + 1):
                gk = groups[k - 1]
                if len(gk):
                    g[0, off:off + len(gk)] = -self.orc.gradK(k, len(gk), gk, invcovs[k - 1], v[0, o].numpy()[None, :])
                off += len(gk)
        return torch.from_numpy(g)

    def combine_grad(self, grad, coef, scale=None, out=None):
        res = np.zeros((1, self.L))
        for o, (K, groups, invcovs, mapping) in enumerate(self.outs):
            res[0, mapping] += float(coef[0, o]) * grad[0, self.grad_off[o]:self.grad_off[o] + len(mapping)].numpy()
        return torch.from_numpy(res if scale is None else res * scale.numpy())


def make_problem():
    """n=7, 3 outputs; output 2 uses a ragged subset of the groups (non-identity mapping)"""
    n, kmax, n_out = 7, 3, 3
    prob = synth.problem(n, kmax, n_out)
    groups = prob["groups"]
    sizes = [len(g) for g in groups]
    rng = np.random.RandomState(3)
    outs = []
    for o in range(n_out):
        if o < 2:
            outs.append({"K": kmax, "sizes": sizes, "groups": groups, "C": prob["C"][o], "mapping": None})
        else:
            keep = [rng.rand(len(g)) < 0.6 for g in groups]
            for kp in keep:
                kp[0] = True
            off, mp = 0, []
            for k, g in enumerate(groups):
                mp.append(off + np.nonzero(keep[k])[0])
                off += len(g)
            outs.append({"K": kmax, "sizes": [int(kp.sum()) for kp in keep], "groups": [g[kp] for g, kp in zip(groups, keep)],
                         "C": prob["C"][o], "mapping": np.concatenate(mp)})
    return prob, sizes, outs


def worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        prob, sizes, outs = make_problem()
        sp = ShardedPlan(prob["n"], sizes, outs, plan_factory=CpuStandInPlan)
        m = prob["m"][0]
        var, grad_local, status = sp.eval(m)
        coef = torch.tensor([[0.2, 0.5, 0.3]], dtype=torch.float64)
        g = sp.global_gradient(grad_local, coef)
        ret[rank] = (var.numpy().copy(), g.numpy().copy(), (sp.lo, sp.hi))
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_eval_world2_gloo(oracle):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), ret), nprocs=world, join=True)
    # unsharded reference through the oracle
    prob, sizes, outs = make_problem()
    full = CpuStandInPlan(prob["n"], sum(sizes), [dict(o, mapping=np.arange(sum(sizes)) if o["mapping"] is None else o["mapping"]) for o in outs])
    m = prob["m"][0]
    var, v, st = full.solve(full.phi(m))
    g = full.combine_grad(full.grad(v, st), torch.tensor([[0.2, 0.5, 0.3]], dtype=torch.float64))
    for r in range(world):
        assert np.abs(ret[r][0] / var.numpy() - 1).max() < 1e-12
        assert np.abs(ret[r][1] - g.numpy()).max() / np.abs(g.numpy()).max() < 1e-12
    assert ret[0][2][1] == ret[1][2][0] and ret[0][2][0] == 0 and ret[1][2][1] == sum(sizes)
    # and the oracle SAP agrees with the stand-in on an identity output (guards the stand-in itself)
    sap = oracle.OracleSAP(prob["C"][0], prob["kmax"], prob["groups"], prob["costs"])
    assert abs(sap.variance(m) / var.numpy()[0, 0] - 1) < 1e-12


def test_shard_partition_properties():
    for n, kmax, world in ((20, 5, 8), (25, 6, 8), (12, 12, 3), (6, 2, 4)):
        sizes = [len(g) for g in synth.all_groups(n, kmax)] if n < 25 else [25, 300, 2300, 12650, 53130, 177100]
        cuts = shard_bounds(sizes, world)
        assert cuts[0] == 0 and cuts[-1] == sum(sizes) and all(b >= a for a, b in zip(cuts, cuts[1:]))
        w = np.concatenate([np.full(s, (k + 1) ** 2) for k, s in enumerate(sizes)])
        loads = [w[a:b].sum() for a, b in zip(cuts, cuts[1:])]
        assert max(loads) <= w.sum() / world + kmax ** 2       # balanced to within one group
    # every group of a ragged output is owned exactly once
    prob, sizes, outs = make_problem()
    cuts = shard_bounds(sizes, 3)
    owned = np.concatenate([shard_output(outs[2], sizes, a, b)["mapping"] for a, b in zip(cuts, cuts[1:])])
    assert sorted(owned.tolist()) == sorted(outs[2]["mapping"].tolist())


class _FakeExchange(object):
    """stand-in for dist.PeerExchange on CPU tensors: all_reduce through the group itself, with failures injected on ONE rank"""

    def __init__(self, max_doubles, device, group=None, fail=None):
        self.rank, self.device, self.fail, self.calls, self.closed = dist.get_rank(group), torch.device("cpu"), fail, 0, False
        self.group = group
        if fail == ("construct", self.rank):
            raise RuntimeError("hipIpcOpenMemHandle: invalid argument (injected)")

    def all_reduce(self, t):
        self.calls += 1
        if self.fail == ("raise", self.rank) and self.calls == 2:
            raise RuntimeError("injected launch failure")
        if self.fail == ("corrupt", self.rank) and self.calls == 3:
            t += 1.0                      # this rank alone disagrees with the reference all_reduce
            return t
        # a correct exchange that does NOT use the group (the real one writes peer memory): every rank can compute the sum
        # because the self-test vectors are seeded
        tot = torch.zeros_like(t)
        for r in range(dist.get_world_size(self.group)):
            tot += _nth_selftest_vector(r, self.calls - 1, self.max_doubles)
        t.copy_(tot)
        return t

    def timed_out(self):
        return self.fail == ("timeout", self.rank) and self.calls >= 4

    def close(self):
        self.closed = True


def _nth_selftest_vector(rank, r_index, max_doubles):
    """the vector PeerExchange.create draws on `rank` in self-test round r_index (same generator, same order)"""
    rng = np.random.RandomState(99 + rank)
    for r in range(r_index + 1):
        n = max_doubles if r % 2 == 0 else max(1, max_doubles // 3)
        a = rng.randn(n) * 10.0 ** rng.randint(-3, 4)
    return torch.from_numpy(a)


def exchange_worker(rank, world, port, ret):
    from bluest_amd.dist import PeerExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        for fail in (None, ("construct", 1), ("raise", 0), ("corrupt", 1), ("timeout", 0)):
            made = []

            def factory(max_doubles, device, group=None, fail=fail):
                ex = _FakeExchange(max_doubles, device, group=group, fail=fail)
                ex.max_doubles = max_doubles
                made.append(ex)
                return ex
            ex = PeerExchange.create(30, "cpu", rounds=6, _factory=factory)
            # the group is still usable and in step after every scenario: a plain collective gives the right answer
            t = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(t)
            out[str(fail)] = (ex is not None, float(t[0]), [e.closed for e in made])
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_peer_exchange_create_survives_one_sided_failures():
    """a rank whose constructor raises, or whose self-test fails alone, must not leave the others inside collectives of
    a different shape.  Every scenario ends with the same decision on both ranks and a group that still works."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(exchange_worker, args=(world, free_port(), ret), nprocs=world, join=True)
    for fail in ("None", "('construct', 1)", "('raise', 0)", "('corrupt', 1)", "('timeout', 0)"):
        a, b = ret[0][fail], ret[1][fail]
        assert a[0] == b[0] == (fail == "None"), (fail, a, b)
        assert a[1] == b[1] == 3.0
        if fail != "None":
            assert all(a[2]) and all(b[2])          # whatever was constructed has been closed on both ranks
