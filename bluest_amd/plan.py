"""
Device-resident evaluation plan: Python face of Part 2 of include/bluest_hip.h.

torch is used only as plumbing: it owns device buffers and the current HIP stream; every number is produced by
the hand-written kernels in csrc/plan.hip (and csrc/spg.hip for the projection).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

EVAL_OK, EVAL_INF, EVAL_NO_MODEL0, EVAL_SINGULAR = 0, 1, 2, 3


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def require_cuda():
    if not torch.cuda.is_available():
        raise _lib.BluestHipError("no GPU visible to torch: bluest_amd computes on MI355X only (no CPU fallback)")


class _LazyInverses(object):
    def __init__(self, plan):
        import weakref
        self._plan_ref, self._cache = weakref.ref(plan), {}      # weak: plan -> this -> plan would be a reference cycle

    def __len__(self): return self._plan_ref().n_out

    def __getitem__(self, o):
        o = int(o)
        if o not in self._cache:
            pl = self._plan_ref()
            ic = np.empty(pl._n_inv[o], dtype=np.float64)
            with torch.cuda.device(pl.device):
                check(pl.lib.bluest_plan_get_invcovs(pl._h, o, ptr(ic)))
            self._cache[o] = ic
        return self._cache[o]


class Plan(object):
    """One plan = all outputs of one (multi-output) sample-allocation problem, resident in HBM.

    outputs: list of dicts with keys
        K        max group size of this output
        sizes    L_k for k = 1..K
        groups   list over k of (L_k, k) int arrays (or one flat array)
        invcovs  list over k of flat (L_k*k*k) float arrays, OR
        C        (N, N) covariance -> per-group pseudo-inverses are computed on the GPU (sap.py:69-79)
        mapping  (L_o,) indices into the global allocation vector, or None for identity
    """

    def __init__(self, n_models, L_global, outputs, max_candidates=1, device=None):
        require_cuda()
        self.lib = _lib.lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.N = int(n_models)
        self.L = int(L_global)
        self.n_out = len(outputs)
        self.max_candidates = int(max_candidates)
        self._h = ctypes.c_void_p()
        self._n_inv = []
        with torch.cuda.device(self.device):
            check(self.lib.bluest_plan_create(ctypes.byref(self._h), self.N, self.L))
            try:
                self._sizes = []
                self._mappings = [out.get("mapping") for out in outputs]      # host view of each output's local -> global map
                for out in outputs:
                    K = int(out["K"])
                    sizes = _i64(out["sizes"])
                    assert len(sizes) == K
                    self._sizes.append(sizes.copy())
                    g = out["groups"]
                    groups = _i64(np.concatenate([np.asarray(x, dtype=np.int64).ravel() for x in g])
                                  if isinstance(g, (list, tuple)) else g)
                    mapping = None if out.get("mapping") is None else _i64(out["mapping"])
                    if out.get("invcovs") is not None:
                        ic = out["invcovs"]
                        ic = _f64(np.concatenate([np.asarray(x, dtype=np.float64).ravel() for x in ic])
                                  if isinstance(ic, (list, tuple)) else ic)
                        check(self.lib.bluest_plan_add_output(self._h, K, ptr(sizes), ptr(groups), ptr(ic), ptr(mapping)))
                    else:
                        C = _f64(out["C"])
                        # the per-group pseudo-inverses are computed AND kept on the device; the host asks for them lazily
                        check(self.lib.bluest_plan_add_output_cov(self._h, ptr(C), K, ptr(sizes), ptr(groups), ptr(mapping), None))
                    self._n_inv.append(int(sum(int(sizes[k - 1]) * k * k for k in range(1, K + 1))))
                check(self.lib.bluest_plan_finalize(self._h, self.max_candidates))
            except Exception:
                self.lib.bluest_plan_destroy(self._h)
                self._h = None
                raise
        self._after_finalize()

    def _after_finalize(self):
        glen = ctypes.c_int64(0)
        offs = (ctypes.c_int64 * self.n_out)()
        check(self.lib.bluest_plan_grad_layout(self._h, ctypes.byref(glen), offs))
        self.grad_len = glen.value
        self.grad_off = [int(x) for x in offs]
        pb, gb = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self.lib.bluest_plan_traffic(self._h, ctypes.byref(pb), ctypes.byref(gb)))
        self.phi_bytes, self.grad_bytes = pb.value, gb.value
        self.reclen = self.N * self.N + 2 * self.N + 1
        mf, mfb = ctypes.c_int(0), ctypes.c_int64(0)
        check(self.lib.bluest_plan_matfree(self._h, ctypes.byref(mf), ctypes.byref(mfb)))
        # single-candidate evaluations recompute the group inverses (csrc/matfree.hip): Phi and gradient (matfree), or the gradient only
        self.matfree, self.matfree_gradient, self.matfree_bytes = mf.value == 1, mf.value in (1, 2), mfb.value
        ident = ctypes.c_int(0)
        check(self.lib.bluest_plan_is_identity(self._h, ctypes.byref(ident)))
        self.identity = ident.value == 1            # every output on all groups, local index = global index

    def restrict(self, keep, max_candidates=1):
        """a new Plan over the sub-list `keep` (sorted global group indices) of this plan's groups: allocation vectors of length
        len(keep), every output keeps its groups inside `keep`, pseudo-inverses gathered on the device (bluest_plan_restrict).
        Raises BLUESTError-compatible RuntimeError (via check) if an output would lose model 0."""
        keep = _i64(keep)
        sub = Plan.__new__(Plan)
        sub.lib, sub.device, sub.N, sub.L, sub.n_out = self.lib, self.device, self.N, len(keep), self.n_out
        sub.max_candidates = int(max_candidates)
        sub._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.bluest_plan_restrict(self._h, ptr(keep), len(keep), sub.max_candidates, ctypes.byref(sub._h)))
        sub._sizes, sub._mappings, sub._n_inv = [], [], []
        for o in range(sub.n_out):
            K = ctypes.c_int(0)
            check(self.lib.bluest_plan_output_layout(sub._h, o, ctypes.byref(K), None, None))
            sizes = np.zeros(K.value, dtype=np.int64)
            check(self.lib.bluest_plan_output_layout(sub._h, o, ctypes.byref(K), ptr(sizes), None))
            mapping = np.zeros(int(sizes.sum()), dtype=np.int64)
            check(self.lib.bluest_plan_output_layout(sub._h, o, ctypes.byref(K), ptr(sizes), ptr(mapping)))
            sub._sizes.append(sizes)
            sub._mappings.append(mapping)
            sub._n_inv.append(int(sum(int(sizes[k - 1]) * k * k for k in range(1, K.value + 1))))
        sub._after_finalize()
        return sub

    @property
    def invcovs(self):
        """list over outputs of the reference-layout pseudo-inverses (flat, concat over group sizes; sap.py:69-79), fetched
        from the device on first use"""
        if getattr(self, "_invcovs", None) is None:
            self._invcovs = _LazyInverses(self)
        return self._invcovs

    def gather_invcovs(self, output, local_idx):
        """k x k blocks of some groups of one output (local indices), concatenated: gathered on the device, one small copy"""
        idx = _i64(local_idx)
        if len(idx) == 0:
            return np.zeros(0)
        sizes = self._group_sizes(output)[idx]
        out = np.empty(int((sizes * sizes).sum()), dtype=np.float64)
        with torch.cuda.device(self.device):
            check(self.lib.bluest_plan_gather_invcovs(self._h, int(output), ptr(idx), len(idx), ptr(out)))
        return out

    def _group_sizes(self, output):
        cache = self.__dict__.setdefault("_gsizes", {})
        if output not in cache:
            cache[output] = np.concatenate([np.full(int(n), k + 1, dtype=np.int64) for k, n in enumerate(self._sizes[output])])
        return cache[output]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.bluest_plan_destroy(self._h)      # switches to the plan's own device for the release (DeviceScope)
                self._h = None
        except Exception:
            pass

    # ---- helpers ------------------------------------------------------------------------------------
    def to_device(self, m):
        """host/device array -> contiguous float64 device tensor of shape (n_cand, L)"""
        if isinstance(m, torch.Tensor):
            t = m.to(device=self.device, dtype=torch.float64)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(m), dtype=np.float64)).to(self.device)
        if t.dim() == 1:
            t = t.unsqueeze(0)
        t = t.contiguous()
        if t.shape[1] != self.L:
            raise ValueError("allocation vector has length %d, plan expects %d" % (t.shape[1], self.L))
        if t.shape[0] > self.max_candidates:
            raise ValueError("%d candidates > max_candidates=%d" % (t.shape[0], self.max_candidates))
        return t

    # ---- fused single-GPU evaluation (MOSAP.variances / variance_GH) ----------------------------------
    def eval(self, m, delta=0.0, want_grad=True, out=None):
        """returns (var (n_cand,n_out), grad (n_cand,grad_len) | None, status (n_cand,n_out) int32) device tensors"""
        m = self.to_device(m)
        nc = m.shape[0]
        if out is None:
            var = torch.empty((nc, self.n_out), dtype=torch.float64, device=self.device)
            grad = torch.empty((nc, self.grad_len), dtype=torch.float64, device=self.device) if want_grad else None
            status = torch.empty((nc, self.n_out), dtype=torch.int32, device=self.device)
        else:
            var, grad, status = out
        with torch.cuda.device(self.device):
            check(self.lib.bluest_plan_eval(self._h, m.data_ptr(), nc, m.stride(0), float(delta), var.data_ptr(),
                                            None if grad is None else grad.data_ptr(),
                                            0 if grad is None else grad.stride(0), status.data_ptr(), _stream()))
        return var, grad, status

    # ---- three-phase path (multi-GPU: all-reduce the record between phi() and solve()) ----------------
    def phi(self, m, out=None):
        m = self.to_device(m)
        nc = m.shape[0]
        rec = out if out is not None else torch.empty((nc, self.n_out, self.reclen), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.bluest_plan_phi(self._h, m.data_ptr(), nc, m.stride(0), rec.data_ptr(), _stream()))
        return rec

    def solve(self, rec, delta=0.0, out=None):
        nc = rec.shape[0]
        if out is None:
            var = torch.empty((nc, self.n_out), dtype=torch.float64, device=self.device)
            v = torch.empty((nc, self.n_out, self.N), dtype=torch.float64, device=self.device)
            status = torch.empty((nc, self.n_out), dtype=torch.int32, device=self.device)
        else:
            var, v, status = out
        with torch.cuda.device(self.device):
            check(self.lib.bluest_plan_solve(self._h, rec.data_ptr(), nc, float(delta), var.data_ptr(), v.data_ptr(),
                                             status.data_ptr(), _stream()))
        return var, v, status

    def eval_pinv(self, m, delta=0.0):
        """the evaluation for a rank-deficient restricted Phi (rare; the caller saw EVAL_SINGULAR): V and grad V as the reference's
        variance_GH computes them through numpy's pinv (misc.py:487,490).  Returns (var, grad, status) like eval()."""
        rec = self.phi(m)
        nc = rec.shape[0]
        var = torch.empty((nc, self.n_out), dtype=torch.float64, device=self.device)
        v = torch.empty((nc, self.n_out, self.N), dtype=torch.float64, device=self.device)
        status = torch.empty((nc, self.n_out), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.bluest_plan_solve_pinv(self._h, rec.data_ptr(), nc, float(delta), var.data_ptr(), v.data_ptr(),
                                                  status.data_ptr(), _stream()))
        return var, self.grad(v, status), status

    def solve_grad(self, rec, delta=0.0, out=None):
        """solve + gradient from an (all-reduced) Phi record of ONE candidate in one launch: (var, grad, status)"""
        assert rec.shape[0] == 1
        if out is None:
            var = torch.empty((1, self.n_out), dtype=torch.float64, device=self.device)
            grad = torch.empty((1, self.grad_len), dtype=torch.float64, device=self.device)
            status = torch.empty((1, self.n_out), dtype=torch.int32, device=self.device)
        else:
            var, grad, status = out
        with torch.cuda.device(self.device):
            check(self.lib.bluest_plan_solve_grad(self._h, rec.data_ptr(), float(delta), var.data_ptr(), grad.data_ptr(),
                                                  status.data_ptr(), None, 0, None, _stream()))
        return var, grad, status

    def grad(self, v, status, out=None):
        nc = v.shape[0]
        grad = out if out is not None else torch.empty((nc, self.grad_len), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            check(self.lib.bluest_plan_grad(self._h, v.data_ptr(), status.data_ptr(), nc, grad.data_ptr(), grad.stride(0), _stream()))
        return grad

    def combine_grad(self, grad, coef, scale=None, out=None):
        """out[c][j] = scale[j] * sum_o coef[c][o] * grad_o[c][local_o(j)]"""
        nc = grad.shape[0]
        res = out if out is not None else torch.empty((nc, self.L), dtype=torch.float64, device=self.device)
        coef = coef.to(device=self.device, dtype=torch.float64).contiguous()
        with torch.cuda.device(self.device):
            check(self.lib.bluest_plan_combine_grad(self._h, grad.data_ptr(), grad.stride(0), coef.data_ptr(),
                                                    None if scale is None else scale.data_ptr(), nc, res.data_ptr(),
                                                    res.stride(0), _stream()))
        return res

    def output_gradients(self, grad):
        """(n_out, L) host array: row o = gradient of V_o scattered to the global allocation vector (zero outside the
        output's groups), from the concatenated per-output gradient of ONE candidate as eval() returns it"""
        gh = np.asarray(grad.cpu().numpy() if isinstance(grad, torch.Tensor) else grad, dtype=np.float64).reshape(-1)
        G = np.zeros((self.n_out, self.L))
        for o in range(self.n_out):
            mp = self._mappings[o]
            if mp is None:
                G[o] = gh[self.grad_off[o]:self.grad_off[o] + self.L]
            else:
                G[o, np.asarray(mp, dtype=np.int64)] = gh[self.grad_off[o]:self.grad_off[o] + len(mp)]
        return G

    def phi_matrix(self, m, delta=0.0):
        """Phi(m) + delta*I for every output as an (n_cand, n_out, N, N) device tensor (misc.py:459-461)"""
        rec = self.phi(m)
        N = self.N
        # re-packed on the host (a few KB): a strided device copy would be the first use of a torch operator in the process,
        # which costs ~100 ms of kernel loading on ROCm
        PHI = np.ascontiguousarray(rec.cpu().numpy()[:, :, :N * N]).reshape(rec.shape[0], self.n_out, N, N)
        if delta:
            PHI = PHI + float(delta) * np.eye(N)
        return torch.from_numpy(PHI).to(self.device)


def simplex_project(x, g=None, lmbda=0.0, z=1.0, want_p=True, want_d=True, floor=0.0):
    """p = P_simplex(x - lmbda*s*g) in the metric diag(1/s), d = p - x, on the GPU; s = 1 (floor = 0) or
    max(x, floor).  Returns (p, d, stats[4] device tensor)"""
    require_cuda()
    L = x.numel()
    p = torch.empty_like(x) if want_p else None
    d = torch.empty_like(x) if want_d else None
    stats = torch.empty(4, dtype=torch.float64, device=x.device)
    work = projection_workspace(L, x.device) if L > 4096 else None
    with torch.cuda.device(x.device):
        check(_lib.lib().bluest_simplex_project(x.data_ptr(), None if g is None else g.data_ptr(), float(lmbda), float(z), float(floor), L,
                                                None if p is None else p.data_ptr(), None if d is None else d.data_ptr(),
                                                stats.data_ptr(), None if work is None else work.data_ptr(), _stream()))
    return p, d, stats


_WORKSPACES = {}


def projection_workspace(L, device):
    """scratch for the multi-CU projection path (cached per (L, device))"""
    key = (int(L), str(device))
    if key not in _WORKSPACES:
        n = ctypes.c_int64(0)
        check(_lib.lib().bluest_simplex_workspace_doubles(int(L), ctypes.byref(n)))
        _WORKSPACES[key] = torch.from_numpy(np.zeros(n.value)).to(device)     # zero-filled once (tag 0 = empty mailbox)
    return _WORKSPACES[key]
