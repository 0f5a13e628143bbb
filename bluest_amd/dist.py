"""
Multi-GPU evaluation: the group set is sharded over the ranks of a torch.distributed process group (one process
per GPU, backend "nccl" = RCCL over xGMI); the only exchange per evaluation is ONE all-reduce(SUM) of the partial
Phi records, n_out*(N*N+2N+1) float64.  Every rank then solves the n x n systems
redundantly and evaluates the gradient of ITS groups; nothing else crosses the fabric.

The reference has no counterpart (its optimiser runs on MPI rank 0 only, bluest/blue_models.py:508-526).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(sizes, world):
    """contiguous cuts of the size-major global group list, balanced by sum k^2 (the streamed bytes).
    sizes[k-1] = number of groups of size k.  Returns world+1 cut points into the global numbering."""
    w = np.concatenate([np.full(int(n), (k + 1) ** 2, dtype=np.float64) for k, n in enumerate(sizes)]) if sum(sizes) else np.zeros(0)
    c = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [int(np.searchsorted(c, c[-1] * r / world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, len(w)
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts


def shard_output(out, L_global_sizes, lo, hi):
    """restrict one output description (see plan.Plan) to the groups whose GLOBAL index (through its mapping) lies
    in [lo, hi).  Works for identity and ragged mappings; keeps the size-major order."""
    K = int(out["K"])
    sizes = [int(x) for x in out["sizes"]]
    groups = out["groups"]
    if not isinstance(groups, (list, tuple)):
        flat, groups, off = np.asarray(groups, dtype=np.int64), [], 0
        for k in range(1, K + 1):
            groups.append(flat[off:off + sizes[k - 1] * k].reshape(-1, k))
            off += sizes[k - 1] * k
    mapping = out.get("mapping")
    if mapping is None:
        mapping = np.arange(sum(sizes), dtype=np.int64)
    mapping = np.asarray(mapping, dtype=np.int64)
    invcovs = out.get("invcovs")
    new = {"K": K, "sizes": [], "groups": [], "mapping": []}
    if invcovs is not None:
        new["invcovs"] = []
    else:
        new["C"] = out["C"]
    off = 0
    for k in range(1, K + 1):
        n = sizes[k - 1]
        mp = mapping[off:off + n]
        keep = (mp >= lo) & (mp < hi)
        gk = np.asarray(groups[k - 1], dtype=np.int64).reshape(-1, k)
        new["sizes"].append(int(keep.sum()))
        new["groups"].append(gk[keep])
        new["mapping"].append(mp[keep])
        if invcovs is not None:
            new["invcovs"].append(np.asarray(invcovs[k - 1], dtype=np.float64).reshape(-1, k * k)[keep].ravel())
        off += n
    new["mapping"] = np.concatenate(new["mapping"]) if len(new["mapping"]) else np.zeros(0, dtype=np.int64)
    return new


class ShardedPlan(object):
    """plan.Plan over this rank's shard of the groups + the all-reduce of the Phi records.

    global_sizes: L_k of the GLOBAL group list (defines the global numbering that mappings refer to).
    plan_factory: callable(n_models, L_global, outputs, max_candidates, device) -> object with phi/solve/grad
                  (default: the HIP plan; tests inject a CPU stand-in to exercise the wiring under gloo).
    """

    def __init__(self, n_models, global_sizes, outputs, max_candidates=1, device=None, group=None, plan_factory=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.L = int(sum(global_sizes))
        self.cuts = shard_bounds(global_sizes, self.world)
        self.lo, self.hi = self.cuts[self.rank], self.cuts[self.rank + 1]
        local = [shard_output(o, global_sizes, self.lo, self.hi) for o in outputs]
        if any(sum(o["sizes"]) == 0 for o in local):
            raise ValueError("rank %d would own no group of some output: too many ranks for this problem" % self.rank)
        self.local_outputs = local
        if plan_factory is None:
            from .plan import Plan
            plan_factory = Plan
        self.plan = plan_factory(n_models, self.L, local, max_candidates=max_candidates, device=device)
        self.n_out = len(outputs)

    def eval(self, m, delta=0.0, want_grad=True, rec=None, out=None):
        """returns (var (n_cand,n_out), grad_local | None, status); grad_local covers this rank's groups only, in the
        plan's concatenated per-output layout (plan.grad_off, local numbering).  out = (var, grad, status) preallocated
        device tensors (the v workspace is kept by this object)."""
        rec = self.plan.phi(m, out=rec)
        if self.world > 1:
            dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=self.group)
        if out is None:
            var, v, status = self.plan.solve(rec, delta)
            grad = self.plan.grad(v, status) if want_grad else None
            return var, grad, status
        var, grad, status = out
        if getattr(self, "_v", None) is None or self._v.shape[0] != rec.shape[0]:
            self._v = torch.empty((rec.shape[0], self.n_out, self.plan.N), dtype=torch.float64, device=rec.device)
        self.plan.solve(rec, delta, out=(var, self._v, status))
        if want_grad:
            self.plan.grad(self._v, status, out=grad)
        return var, grad if want_grad else None, status

    def global_gradient(self, grad_local, coef, scale=None):
        """g[j] = scale[j] * sum_o coef[o] * dV_o/dm_j for ALL j: every rank fills the entries of its groups, one
        all-reduce(SUM) assembles the vector (entries owned by nobody stay 0)."""
        g = self.plan.combine_grad(grad_local, coef, scale=scale)
        if self.world > 1:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        return g
