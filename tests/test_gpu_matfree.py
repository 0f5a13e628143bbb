"""
GPU tests (-m gpu) of the MATRIX-FREE evaluation (csrc/matfree.hip; BASELINE.json's north-star form of the path: the inverse of
every group's covariance block, bluest/sap.py:69-79, recomputed in registers where bluest/cmisc.cpp:25-40,58-72 stream the stored
one): against the reference's golden vectors, against the stored-inverse path of the same library on the same inputs, for
bit-reproducibility, and for the plans it must refuse.
"""
import os
import sys

import numpy as np
import pytest

from bluest_amd import synth
from conftest import golden, rel_err

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _plan(prob, matfree, monkeypatch, outputs=None, max_candidates=1):
    from bluest_amd.plan import Plan
    monkeypatch.setenv("BLUEST_MATFREE", "1" if matfree else "0")
    return Plan(prob["n"], prob["K_tot"], outputs if outputs is not None else bench.build_outputs(prob), max_candidates=max_candidates)


@pytest.mark.parametrize("fname,via_mosap", [("sap_n5_all.npz", False), ("sap_n6_all.npz", False), ("sap_n20_k5_o8.npz", True)])
def test_matrix_free_path_against_the_reference_golden_vectors(gpu, monkeypatch, fname, via_mosap):
    """Phi, V, grad V of the matrix-free path against the vectors recorded from the real reference (oracle/gen_golden.py): the same
    checks, at the same 1e-11, as the stored-inverse path takes in tests/test_gpu_parity.py -- sparse allocations, a model dropped
    out, only three models sampled, delta > 0, int64 m, all-tiny m -> inf"""
    import test_gpu_parity as tp
    from bluest_amd.plan import Plan
    monkeypatch.setenv("BLUEST_MATFREE", "1")
    made = []
    orig = Plan._after_finalize

    def spy(self):
        orig(self)
        made.append(self.matfree)
    monkeypatch.setattr(Plan, "_after_finalize", spy)
    tp._check_sap_file(fname, via_mosap=via_mosap)
    assert made and all(made), "the plans of this test must evaluate matrix-free"


@pytest.mark.parametrize("n,kmax,n_out", [(8, 3, 1), (16, 4, 3), (20, 5, 8), (12, 8, 2), (25, 6, 1), (40, 2, 2)])
def test_matrix_free_equals_the_stored_inverse_path(gpu, monkeypatch, n, kmax, n_out):
    """same inputs, both paths of the library: V and grad V to 1e-10 (relative to the largest entry), identical status codes, the Phi
    record to 1e-12 -- for dense, sparse (models dropping out) and tiny allocations, delta > 0, and at K_tot = 245 505"""
    torch = gpu
    prob = synth.problem(n, kmax, n_out)
    L = prob["K_tot"]
    pm, ps = _plan(prob, True, monkeypatch), _plan(prob, False, monkeypatch)
    assert pm.matfree and not ps.matfree
    rng = np.random.RandomState(3)
    sparse = np.where(rng.rand(L) < 0.02, 10.0 * rng.rand(L), 0.0)
    sparse[0] = 1.0                                             # group {0}: model 0 stays sampled
    no0 = sparse.copy()
    no0[np.asarray([0 in g for gk in prob["groups"] for g in gk])] = 0.0      # model 0 unsampled: status NO_MODEL0 on both
    few = np.zeros(L); few[:3] = (2.0, 3.0, 0.5)
    for name, m, delta in (("dense", prob["m"][0], 0.0), ("dense+delta", prob["m"][0], 1e-6), ("sparse", sparse, 0.0),
                           ("model 0 out", no0, 0.0), ("few", few, 0.0), ("tiny", np.full(L, 0.01), 0.0)):
        rm, rs = pm.phi(m), ps.phi(m)
        assert rel_err(rm.cpu().numpy(), rs.cpu().numpy()) < 1e-12, name
        vm, gm, sm = pm.eval(m, delta=delta)
        vs, gs, ss = ps.eval(m, delta=delta)
        assert torch.equal(sm, ss), (name, sm, ss)
        ok = (ss[0] == 0).cpu().numpy()
        if ok.any():
            assert rel_err(vm[0].cpu().numpy()[ok], vs[0].cpu().numpy()[ok]) < 1e-10, name
        gm, gs = gm[0].cpu().numpy(), gs[0].cpu().numpy()
        fin = np.isfinite(gs)
        assert (np.isfinite(gm) == fin).all(), name
        if fin.any():
            assert rel_err(gm[fin], gs[fin]) < 1e-10, name
        vm2, _, sm2 = pm.eval(m, delta=delta, want_grad=False)
        assert torch.equal(sm2, sm) and torch.equal(vm2[0][torch.from_numpy(ok).to(vm2.device)], vm[0][torch.from_numpy(ok).to(vm.device)]), name


def test_matrix_free_with_ragged_outputs(gpu, monkeypatch):
    """every output on its own group set (non-identity mappings, bluest/mosap.py:54-67): the matrix-free pass gathers m through the
    output's mapping and writes its gradient in the output's own numbering"""
    from bluest_amd.plan import Plan
    G = golden("mosap_n6_o3_ragged.npz")
    n, n_out, kmax = int(G["n"]), int(G["n_out"]), int(G["kmax"])
    prob = synth.problem(n, kmax, n_out)
    groups = [G["g_k%d" % k] for k in range(1, kmax + 1)]
    flat = {tuple(int(x) for x in g): i for i, g in enumerate(g for gk in groups for g in gk)}
    outs = []
    for o in range(n_out):
        mg = [G["mg%d_k%d" % (o, k)] for k in range(1, kmax + 1)]
        mapping = np.array([flat[tuple(int(x) for x in g)] for gk in mg for g in gk], dtype=np.int64)
        outs.append({"K": kmax, "sizes": [len(g) for g in mg], "groups": mg, "C": prob["C"][o], "mapping": mapping})
    L = len(flat)
    monkeypatch.setenv("BLUEST_MATFREE", "1")
    pm = Plan(n, L, outs)
    monkeypatch.setenv("BLUEST_MATFREE", "0")
    ps = Plan(n, L, outs)
    assert pm.matfree and not ps.matfree
    m = 10.0 * np.random.RandomState(8).rand(L)
    vm, gm, sm = pm.eval(m)
    vs, gs, ss = ps.eval(m)
    assert gpu.equal(sm, ss) and rel_err(vm.cpu().numpy(), vs.cpu().numpy()) < 1e-11 and rel_err(gm.cpu().numpy(), gs.cpu().numpy()) < 1e-11


def test_matrix_free_is_bit_reproducible(gpu, monkeypatch):
    """the scatter-adds stay inside a wavefront's own LDS accumulator and every later sum has a fixed order: two evaluations, and two
    plans, give the same bits (what the redundant solves of a sharded evaluation rely on)"""
    torch = gpu
    prob = synth.problem(20, 5, 8)
    p1, p2 = _plan(prob, True, monkeypatch), _plan(prob, True, monkeypatch)
    assert p1.matfree and p2.matfree
    m = prob["m"][0]
    v0, g0, s0 = p1.eval(m)
    r0 = p1.phi(m)
    for p in (p1, p1, p2):
        v, g, s = p.eval(m)
        assert torch.equal(v, v0) and torch.equal(g, g0) and torch.equal(s, s0) and torch.equal(p.phi(m), r0)


def test_matrix_free_refuses_what_it_cannot_reproduce(gpu, monkeypatch):
    """forced on, the plan still takes the stored pseudo-inverses when a block is (nearly) singular -- the reference's SVD pinv with
    its 1e-15 cut-off (bluest/sap.py:74) is not a Cholesky inverse there --, when an output came with explicit inverses, and when a
    group has more than 8 models; batches of allocation vectors run on the stored path of a matrix-free plan"""
    from bluest_amd.plan import Plan
    from bluest_amd.sap import SAP
    monkeypatch.setenv("BLUEST_MATFREE", "1")
    G = golden("singular_phi_known_answer.npz")
    sap = SAP(G["C"].copy(), 3, [G["g_k%d" % k].copy() for k in (1, 2, 3)], np.ones(8), verbose=False)
    assert not sap.plan.matfree
    Vgh, grad, _ = sap.variance_GH(G["m"], nohess=True)
    assert abs(Vgh / float(G["Vgh_pinv"]) - 1) < 1e-9
    prob = synth.problem(12, 12, 1)
    assert not Plan(12, prob["K_tot"], bench.build_outputs(prob)).matfree                  # groups of up to 12 models
    prob = synth.problem(8, 3, 2)
    pc = Plan(8, prob["K_tot"], bench.build_outputs(prob), max_candidates=4)
    assert pc.matfree
    explicit = [{"K": 3, "sizes": [len(g) for g in prob["groups"]], "groups": prob["groups"], "invcovs": pc.invcovs[o], "mapping": None}
                for o in range(2)]
    assert not Plan(8, prob["K_tot"], explicit).matfree
    ms = 10.0 * np.random.RandomState(2).rand(4, prob["K_tot"])
    vb, gb, sb = pc.eval(ms)                                                               # stored path (batch)
    for c in range(4):
        v1, g1, s1 = pc.eval(ms[c])                                                        # matrix-free (single vector)
        assert rel_err(v1[0].cpu().numpy(), vb[c].cpu().numpy()) < 1e-11 and rel_err(g1[0].cpu().numpy(), gb[c].cpu().numpy()) < 1e-11
