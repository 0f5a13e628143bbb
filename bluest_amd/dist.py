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

from .host import in_host_section


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


class PeerExchange(object):
    """all-reduce(SUM) of a small float64 device buffer by direct peer writes (C-ABI Part 5, csrc/xchg.hip): one kernel launch
    per rank instead of a ring collective -- the Phi record is 5-27 KB, so the exchange is latency-bound.
    The mailbox handles travel through the torch.distributed group once, at construction.  `PeerExchange.create` returns None --
    after telling why on stderr -- when the mailboxes cannot be set up or a self-test against the group's own all_reduce does not
    reproduce it bit for bit on every rank; callers then keep using the group's all_reduce (RCCL)."""

    def __init__(self, max_doubles, device, group=None):
        import ctypes
        from . import _lib
        self.lib, self.group = _lib.lib(), group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device(device)
        self._h = ctypes.c_void_p()
        handle = ctypes.create_string_buffer(64)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.bluest_xchg_create(ctypes.byref(self._h), self.world, self.rank, int(max_doubles), handle))
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(handle.raw), group=group)
            _lib.check(self.lib.bluest_xchg_connect(self._h, b"".join(handles)))
        self.max_doubles = int(max_doubles)

    def all_reduce(self, t):
        """in place; t: contiguous float64 device tensor with at most max_doubles entries"""
        from . import _lib
        from .plan import _stream
        assert t.is_contiguous() and t.dtype == torch.float64 and t.numel() <= self.max_doubles
        with torch.cuda.device(self.device):
            _lib.check(self.lib.bluest_xchg_allreduce_sum(self._h, t.data_ptr(), t.numel(), _stream()))
        return t

    def timed_out(self):
        import ctypes
        flag = ctypes.c_int(0)
        self.lib.bluest_xchg_status(self._h, None, ctypes.byref(flag))
        return bool(flag.value)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.bluest_xchg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def create(max_doubles, device, group=None, rounds=8, _factory=None):
        """collective.  The sequence of collectives is the same on every rank whatever fails where: (1) construction (its
        all_gather of the handles is inside), (2) ONE all-reduce(MIN) agreeing that every rank constructed, (3) -- only if all
        did -- `rounds` self-test rounds, each with the group's all_reduce of the SAME element count on every rank and no early
        exit, (4) ONE all-reduce(MIN) of the local verdicts.  A failure on one rank alone (hipIpcOpenMemHandle, a mailbox
        time-out) therefore ends with `None` on ALL ranks instead of mismatched collectives.
        _factory (tests): callable(max_doubles, device, group) -> exchange object, to inject one-sided failures."""
        import sys
        on_device = dist.get_backend(group) == "nccl"

        def agree(ok):
            flag = torch.tensor([float(ok)], dtype=torch.float64, device=torch.device(device) if on_device else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return float(flag[0]) >= 1.0

        def give_up(ex, why):
            if why:
                sys.stderr.write("bluest_amd.dist: peer-write exchange unavailable on rank %d (%s); using the group's all_reduce\n"
                                 % (dist.get_rank(group), why))
            if ex is not None:
                ex.close()
            return None

        ok, why, ex = 1, "", None
        try:
            ex = (_factory or PeerExchange)(max_doubles, device, group=group)
        except Exception as err:
            ok, why = 0, "%s: %s" % (type(err).__name__, err)
        if not agree(ok):                                          # (2) somebody could not even set up its mailboxes
            return give_up(ex, why or "another rank could not set up its mailboxes")
        rng = np.random.RandomState(99 + dist.get_rank(group))
        buf_dev = torch.device(device) if on_device or ex.device.type == "cuda" else torch.device("cpu")
        for r in range(rounds):                                    # (3) self-test on THIS hardware against the group's all_reduce
            n = max_doubles if r % 2 == 0 else max(1, max_doubles // 3)
            a = torch.from_numpy(rng.randn(n) * 10.0 ** rng.randint(-3, 4)).to(buf_dev)
            b = a.clone()
            try:
                if ok:
                    ex.all_reduce(a)
            except Exception as err:
                ok, why = 0, "%s: %s" % (type(err).__name__, err)
            dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)   # unconditional: same op and count on every rank
            if ok:
                if buf_dev.type == "cuda":
                    torch.cuda.synchronize(buf_dev)
                if ex.timed_out() or not bool(torch.isfinite(a).all()) or float((a - b).abs().max()) > 1e-12 * float(b.abs().max() + 1e-300):
                    ok, why = 0, "self-test round %d disagrees with all_reduce" % r
        if not agree(ok):                                          # (4)
            return give_up(ex, why)
        return ex


class ShardedPlan(object):
    """plan.Plan over this rank's shard of the groups + the all-reduce of the Phi records.

    global_sizes: L_k of the GLOBAL group list (defines the global numbering that mappings refer to).
    plan_factory: callable(n_models, L_global, outputs, max_candidates, device) -> object with phi/solve/grad
                  (default: the HIP plan; tests inject a CPU stand-in to exercise the wiring under gloo).
    """

    def __init__(self, n_models, global_sizes, outputs, max_candidates=1, device=None, group=None, plan_factory=None,
                 exchange="auto"):
        """exchange: "auto" = the one-shot peer-write all-reduce (PeerExchange) when more than one rank runs on GPUs and its
        self-test passes, else the group's all_reduce (RCCL); "collective" = always the group's all_reduce"""
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
        self.N = int(n_models)
        self.device = getattr(self.plan, "device", None)
        self.exchange = None
        if exchange == "auto" and self.world > 1 and plan_factory.__name__ == "Plan" and self.device is not None:
            reclen = self.N * self.N + 2 * self.N + 1
            self.exchange = PeerExchange.create(max_candidates * self.n_out * reclen, self.device, group=group)
        self.exchange_name = "peer-write (bluest_xchg)" if self.exchange is not None else "all_reduce (%s)" % (
            dist.get_backend(group) if dist.is_initialized() else "none")

    def eval(self, m, delta=0.0, want_grad=True, rec=None, out=None):
        """returns (var (n_cand,n_out), grad_local | None, status); grad_local covers this rank's groups only, in the
        plan's concatenated per-output layout (plan.grad_off, local numbering).  out = (var, grad, status) preallocated
        device tensors (the v workspace is kept by this object)."""
        rec = self.plan.phi(m, out=rec)
        self.reduce_records(rec)
        if want_grad and rec.shape[0] == 1 and hasattr(self.plan, "solve_grad"):
            return self.plan.solve_grad(rec, delta, out=out)          # solve + gradient of the shard: one launch
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

    def reduce_records(self, rec):
        """in-place all-reduce(SUM) of Phi records on the current stream: every rank ends with the same bits"""
        if self.world > 1:
            if self.exchange is not None:
                self.exchange.all_reduce(rec)
            else:
                dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=self.group)
        return rec

    @property
    def lib(self):
        return self.plan.lib

    def output_gradients(self, grad_local):
        """plan.Plan.output_gradients for the whole group set: (n_out, K_tot) host array, every rank contributes the rows of its
        groups and one all-reduce(SUM) assembles them (used by the working-set pricing of the solver)"""
        G = torch.from_numpy(self.plan.output_gradients(grad_local))
        if self.world > 1:
            Gd = G.to(self.device) if (self.device is not None and dist.get_backend(self.group) == "nccl") else G
            dist.all_reduce(Gd, op=dist.ReduceOp.SUM, group=self.group)
            G = Gd.cpu()
        return G.numpy()

    def replicated_subplan(self, keep):
        """the (small) plan of the WHOLE problem restricted to the global groups `keep`, built identically on every rank: each rank
        gathers the k x k pseudo-inverse blocks of its own groups inside `keep` on its device, the pieces are all-gathered, and
        every rank assembles the same plan.Plan -- the working set of the solver over a sharded group set (collective)"""
        from .plan import Plan
        from .sap import BLUESTError
        keep = np.asarray(keep, dtype=np.int64)
        if self.world > 1:
            # every rank must ask for the same set (it does, by construction): checked collectively so that a disagreement is an
            # error on ALL ranks rather than a hang in the gather below
            import zlib
            h = float(zlib.crc32(keep.tobytes()))
            t = torch.tensor([h, -h, float(len(keep)), -float(len(keep))], dtype=torch.float64)
            if self.device is not None and dist.get_backend(self.group) == "nccl":
                t = t.to(self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            t = t.cpu().numpy()
            if t[0] != -t[1] or t[2] != -t[3]:
                raise BLUESTError("sharded working set: the ranks ask for different group sets")
        mine = []
        for o, out in enumerate(self.local_outputs):
            mp = np.asarray(out["mapping"], dtype=np.int64)
            local = np.flatnonzero(np.isin(mp, keep))
            cum = np.cumsum([0] + [int(x) for x in out["sizes"]])
            ks = np.searchsorted(cum, local, side="right")                       # group size of every kept local group
            groups = [np.asarray(out["groups"][k - 1], dtype=np.int64).reshape(-1, k)[local[ks == k] - cum[k - 1]]
                      for k in range(1, int(out["K"]) + 1)]
            blocks = self.plan.gather_invcovs(o, local) if len(local) else np.zeros(0)   # concatenated in the order of `local`
            mine.append((mp[local], ks, groups, blocks))
        pieces = [mine]
        if self.world > 1:
            pieces = [None] * self.world
            dist.all_gather_object(pieces, mine, group=self.group)
        outs = []
        for o in range(self.n_out):
            K = int(self.local_outputs[o]["K"])
            gidx = [[] for _ in range(K)]
            grp = [[] for _ in range(K)]
            blk = [[] for _ in range(K)]
            for part in pieces:                                                   # rank order = ascending global index
                idx, ks, groups, blocks = part[o]
                off = 0
                pos = [0] * K
                for i in range(len(idx)):
                    k = int(ks[i])
                    gidx[k - 1].append(int(idx[i]))
                    grp[k - 1].append(groups[k - 1][pos[k - 1]])
                    blk[k - 1].append(blocks[off:off + k * k])
                    pos[k - 1] += 1
                    off += k * k
            sizes, groups_k, inv_k, glob = [], [], [], []
            for k in range(1, K + 1):
                order = np.argsort(np.asarray(gidx[k - 1], dtype=np.int64), kind="stable")
                sizes.append(len(order))
                groups_k.append(np.asarray([grp[k - 1][j] for j in order], dtype=np.int64).reshape(-1, k))
                inv_k.append(np.concatenate([blk[k - 1][j] for j in order]) if len(order) else np.zeros(0))
                glob.extend(gidx[k - 1][j] for j in order)
            if not any(len(g) and (g == 0).any() for g in groups_k):
                raise BLUESTError("restricted plan: output %d would not sample model 0" % o)
            outs.append({"K": K, "sizes": sizes, "groups": groups_k, "invcovs": inv_k,
                         "mapping": np.searchsorted(keep, np.asarray(glob, dtype=np.int64))})
        return Plan(self.N, len(keep), outs, max_candidates=1, device=self.device)

    def combine_grad(self, grad_local, coef, scale=None, out=None):
        """plan.Plan.combine_grad for the whole group set (same signature): see global_gradient"""
        return self.global_gradient(grad_local, coef, scale=scale)

    def global_gradient(self, grad_local, coef, scale=None):
        """g[j] = scale[j] * sum_o coef[o] * dV_o/dm_j for ALL j: every rank fills the entries of its groups, one
        all-reduce(SUM) assembles the vector (entries owned by nobody stay 0)."""
        g = self.plan.combine_grad(grad_local, coef, scale=scale)
        if self.world > 1:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
        return g


@in_host_section
def sharded_spg(sharded, costs, budget=None, eps=None, x0=None, params=None):
    """solver="spg" over a group-sharded plan (COLLECTIVE).  On GPUs: the second-order finish of bluest_amd/colgen.py -- the
    multiplicative phase keeps the allocation, its gradient and the iterate SHARDED (each rank updates the entries of its own
    groups from the all-reduced Phi record; nothing of length K_tot crosses the fabric), the column generation gathers <= 1024
    pricing candidates and the blocks of the <= 64 support groups per round, and every rank solves the small master problem
    redundantly with the same deterministic kernel: identical bits on all ranks.  params={"method": "spg"} selects the
    first-order loop on replicated vectors (spg_device.ShardedDeviceSpg: one more all-reduce of the K_tot gradient per step);
    CPU stand-in plans (tests under gloo) use the host-driven driver bluest_amd.spg.spg = bluest/spg.py:39-132.
    The reference runs its optimiser on one MPI rank (bluest/blue_models.py:508-526); this is the N-GPU counterpart of
    SAP.solve / MOSAP.solve(..., continuous_relaxation=True).  Returns (samples, solver info)."""
    from .sap import SpgAllocator
    on_gpu = sharded.device is not None and type(sharded.plan).__name__ == "Plan"
    prm = {"device_loop": on_gpu, "eps": 1.0e-7}
    if not on_gpu:
        prm["maxit"] = 600
    if params:
        prm.update(params)
    if not on_gpu:
        prm["device_loop"] = False                                 # CPU stand-in plans (tests): the host-driven driver
        prm["method"] = "spg"
    alloc = SpgAllocator(sharded, costs, None, verbose=False)
    m = alloc.solve(budget=budget, eps=eps, x0=x0, params=prm)
    return m, alloc.info
