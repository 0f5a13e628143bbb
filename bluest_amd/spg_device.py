"""
Device-resident SPG: the iteration of bluest/spg.py:68-106 (direction by projection, nonmonotone Armijo line search with
safeguarded quadratic interpolation, Barzilai-Borwein step) with ALL control flow on the GPU.  The solver state is a
256-double array in HBM; the loop is a sequence of identical STEPS, each a fixed sequence of kernels: direction (or, while a
line search is pending, the next trial point), T predicated line-search slots (Phi pass + solve + decision), gradient, update.
A trial rejected in the last slot leaves the line search pending and the next step continues it, so nothing ever waits for
the host, which only looks at the state every `check_every` steps (convergence test, failure).

Same algorithm as bluest_amd.spg.spg with SpgAllocator's callbacks (tests/test_gpu_api.py compares the two
iteration by iteration); this is what `solve(..., solver="spg")` runs by default.
"""
import ctypes
import time

import numpy as np
import torch

from ._lib import capture_guard, check
from .plan import EVAL_OK, _stream, projection_workspace, simplex_project

# state layout (csrc/spg.hip SPG_*)
F, FNEW, LAMBDA, ALPHA, GD, DMAX, TAU, NPOS, ACCEPT, FAIL, DONE, IT, COUNT, NORM, P, LMIN, LMAX, HLEN, SDOTS, SDOTY, FTRIAL = range(21)
EPS, PENDING, MAXFEV, GPSTATS = 21, 22, 23, 24
HIST, COEF, S, STATE_DOUBLES = 32, 64, 128, 256


class DeviceSpg(object):
    def __init__(self, plan, scale, s_norm, p, floor, lmbda_min=1e-30, lmbda_max=1e30, Hlength=10, slots=2, check_every=20):
        assert 1 <= Hlength <= 16 and plan.n_out <= 64
        # self.plan answers eval()/combine_grad() for the WHOLE group set; self.hip is the HIP plan whose handle the launch
        # sequences use (the same object, except for a group-sharded plan: its local shard)
        self.hip = getattr(plan, "plan", plan)
        self.plan, self.lib, self.dev = plan, self.hip.lib, self.hip.device
        self.L, self.n_out = plan.L, plan.n_out
        self.scale = scale.contiguous()
        self.scale_h = self.scale.cpu().numpy()
        # NOTE: no torch compute operator is used anywhere on this path (only allocations and copies): on ROCm the first use
        # of each torch operator loads its kernels, 30-150 ms a piece, which used to triple the first solve of a process
        self._one = torch.from_numpy(np.ones(1)).to(self.dev)
        self._zero = torch.from_numpy(np.zeros(1)).to(self.dev)
        self.s_norm = np.asarray(s_norm, dtype=np.float64)
        self.p, self.floor = float(p), float(floor)
        self.lmin, self.lmax, self.H = float(lmbda_min), float(lmbda_max), int(Hlength)
        self.T, self.check_every = int(slots), int(check_every)
        d = dict(dtype=torch.float64, device=self.dev)
        L = self.L
        self.x, self.g, self.d = torch.empty(L, **d), torch.empty(L, **d), torch.empty(L, **d)
        self.xnew, self.gnew, self.m = torch.empty(L, **d), torch.empty(L, **d), torch.empty(L, **d)
        self.var = torch.empty((1, self.n_out), **d)
        self.status = torch.from_numpy(np.zeros((1, self.n_out), dtype=np.int32)).to(self.dev)
        self.grad = torch.empty((1, self.hip.grad_len), **d)
        self.enable = torch.from_numpy(np.ones(1, dtype=np.int32)).to(self.dev)
        self.st = torch.from_numpy(np.zeros(STATE_DOUBLES)).to(self.dev)
        self.work = torch.from_numpy(np.zeros(1024)).to(self.dev)
        self.pws = projection_workspace(L, self.dev)
        v = ctypes.c_void_p()
        check(self.lib.bluest_plan_v_workspace(self.hip._h, ctypes.byref(v), None))
        self.v_ws = v.value
        self.graph_sets = {}          # hipGraphs per number of in-iteration line-search slots
        self.window_seconds = []
        self.graphs = None

    # ---- launch sequences (captured into hipGraphs) -------------------------------------------------------------
    def _direction(self):
        """d = P_s(x - lambda s g) - x and, fused, the first trial point (alpha = 1) + open gate"""
        check(self.lib.bluest_spg_direction(self.x.data_ptr(), self.g.data_ptr(), self.st.data_ptr(), 1.0, self.floor, self.L,
                                            self.d.data_ptr(), self.scale.data_ptr(), self.xnew.data_ptr(), self.m.data_ptr(),
                                            self.enable.data_ptr(), self.pws.data_ptr(), _stream()))

    def _slot(self, t, with_trial):
        if with_trial:
            check(self.lib.bluest_spg_trial(self.x.data_ptr(), self.d.data_ptr(), self.scale.data_ptr(), self.st.data_ptr(),
                                            self.xnew.data_ptr(), self.m.data_ptr(), self.enable.data_ptr(), self.L, _stream()))
        # Phi pass, then solve with the line-search decision fused into its tail (csrc/plan.hip: k_solve_from_chunks)
        check(self.lib.bluest_plan_eval_decide(self.hip._h, self.m.data_ptr(), 0.0, self.var.data_ptr(), self.status.data_ptr(),
                                               self.st.data_ptr(), 1 if t == self.T - 1 else 0, self.enable.data_ptr(), _stream()))

    def _finish(self):
        check(self.lib.bluest_spg_finish(self.hip._h, self.v_ws, self.status.data_ptr(), self.x.data_ptr(), self.g.data_ptr(),
                                         self.xnew.data_ptr(), self.grad.data_ptr(), self.scale.data_ptr(), self.st.data_ptr(),
                                         self.floor, self.work.data_ptr(), _stream()))

    def _converged(self):
        check(self.lib.bluest_spg_converged(self.x.data_ptr(), self.g.data_ptr(), self.st.data_ptr(), 1.0, self.floor, self.L,
                                            self.pws.data_ptr(), _stream()))

    def _iteration(self):
        self._direction()
        for t in range(self.T):
            self._slot(t, t > 0)
        self._finish()

    def _iteration_checked(self):
        self._iteration()
        self._converged()

    def _window_direct(self, n_iterations, check_last):
        """n whole iterations (+ the convergence projection) enqueued by ONE call into the library: plain stream launches"""
        check(self.lib.bluest_spg_window(self.hip._h, self.x.data_ptr(), self.g.data_ptr(), self.d.data_ptr(), self.xnew.data_ptr(),
                                         self.m.data_ptr(), self.scale.data_ptr(), self.st.data_ptr(), self.var.data_ptr(),
                                         self.status.data_ptr(), self.grad.data_ptr(), self.enable.data_ptr(), self.work.data_ptr(),
                                         self.pws.data_ptr(), self.v_ws, self.floor, self.T, int(n_iterations), 1 if check_last else 0,
                                         _stream()))

    def _window(self):
        """check_every iterations, the last one with the convergence projection: ONE graph replay per host look
        (a replay per iteration costs ~8 us more per iteration in launch overhead, tools/spg_floor.py)"""
        for _ in range(self.check_every - 1):
            self._iteration()
        self._iteration_checked()

    def _capture(self, fn):
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            saved = torch.empty_like(self.st)
            saved.copy_(self.st)
            self.st[DONE:DONE + 1].copy_(self._one)   # warm-up outside capture with every kernel predicated off
            fn()
            self.st.copy_(saved)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        # Anything that frees device memory or destroys a hipGraph on this thread while the capture runs invalidates it
        # (hipErrorStreamCaptureInvalidated).  Plans and solver graphs dropped meanwhile -- by the cyclic collector OR by a
        # reference count reaching zero -- are parked by the library / by capture_guard and released after the capture.
        with capture_guard():
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                fn()
        return g

    def __del__(self):
        try:
            graphs = [g for gs in self.graph_sets.values() for g in gs.values()]
            self.graph_sets = {}
            capture_guard.park(graphs)
        except Exception:
            pass

    def _check_replicas(self, hs):
        """hook of the collective variant: all ranks must have seen the same state (no-op on one GPU)"""
        return None

    # ---- host-side objective (initialisation only) ---------------------------------------------------------------
    def _objective(self, var):
        r = var / self.s_norm
        if not np.isfinite(r).all():
            return np.inf, None
        rmax = r.max()
        if np.isinf(self.p) or self.n_out == 1:
            coef = np.zeros(self.n_out)
            coef[int(np.argmax(r))] = 1.0
            return rmax, coef / self.s_norm
        t = (r / rmax) ** self.p
        return rmax * t.sum() ** (1.0 / self.p), (r / rmax) ** (self.p - 1) * t.sum() ** (1.0 / self.p - 1.0) / self.s_norm

    def run(self, x0, eps=1e-7, maxit=2000, max_fevals=10 ** 6, use_graph=None, rel_tol=0.0, stall_window=100):
        """use_graph: None = the default (direct launches of a whole window by one library call; BLUEST_SPG_GRAPH=1 in the
        environment selects hipGraph replay), True = captured hipGraphs, False = direct launches"""
        plan, lib, st = self.plan, self.lib, self.st
        del self.window_seconds[:]                                  # diagnostics of THIS run only
        if use_graph is None:
            import os
            use_graph = bool(os.environ.get("BLUEST_SPG_GRAPH"))
        with torch.cuda.device(self.dev):
            check(lib.bluest_plan_set_gate(self.hip._h, None, 0))
            if not isinstance(x0, torch.Tensor):
                x0 = torch.from_numpy(np.ascontiguousarray(x0, dtype=np.float64))
            self.x.copy_(simplex_project(x0.to(self.dev), want_d=False)[0])
            var, grad, status = plan.eval(self.scale_h * self.x.cpu().numpy())
            if not (status[0].cpu().numpy() == EVAL_OK).all():
                raise RuntimeError("device SPG: infeasible starting point")
            F0, coef = self._objective(var[0].cpu().numpy())
            norm = F0
            self.g.copy_(plan.combine_grad(grad, torch.from_numpy(coef / norm).to(self.dev).reshape(1, -1), scale=self.scale)[0])
            _, _, stats = simplex_project(self.x, self.g, 1.0, want_p=False, floor=self.floor)
            gpmax = float(stats[1])
            h = np.zeros(STATE_DOUBLES)
            h[F] = 1.0
            h[LAMBDA] = min(self.lmax, max(self.lmin, 1.0 / gpmax)) if gpmax > 1e-15 else 0.0
            h[ALPHA] = 1.0
            h[COUNT] = 1.0
            h[NORM], h[P], h[LMIN], h[LMAX], h[HLEN] = norm, self.p, self.lmin, self.lmax, self.H
            h[HIST:HIST + 16] = -np.inf
            h[HIST] = 1.0
            h[S:S + self.n_out] = self.s_norm
            h[EPS] = eps
            h[MAXFEV] = float(max_fevals)
            h[GPSTATS + 1] = gpmax
            st.copy_(torch.from_numpy(h))
            check(lib.bluest_plan_set_gate(self.hip._h, self.enable.data_ptr(), 1))
            try:
                def bind():
                    """launchers of one step, one step + convergence projection, and a whole window; every hipGraph is captured
                    when it is first needed"""
                    if not use_graph:
                        return (lambda: self._window_direct(1, False), lambda: self._window_direct(1, True),
                                lambda: self._window_direct(self.check_every, True))
                    gs = self.graph_sets.setdefault(self.T, {})

                    def lazy(name, fn):
                        def replay():
                            if name not in gs:
                                gs[name] = self._capture(fn)
                            gs[name].replay()
                        return replay
                    self.graphs = gs
                    return lazy("iteration", self._iteration), lazy("iteration_checked", self._iteration_checked), lazy("window", self._window)

                run_iter, run_iter_checked, run_window = bind()
                info, it, count = 1, 0, 1
                stalled = False
                hs = h
                trace = [(0, 1.0)]                                  # (iteration, normalised objective) at the host checks
                checks = [(0, 1, 1.0)]                              # (iteration, evaluations, normalised objective)
                while True:
                    if hs[DONE] != 0.0 or gpmax <= eps:
                        info = 0
                        break
                    # objective stall: the scaled projected-gradient norm does not vanish on this degenerate problem (many
                    # allocations share the optimal variance), so also stop when f stopped moving over `stall_window` iterations
                    old = [f for (i, f) in trace if i <= it - stall_window]
                    if rel_tol > 0.0 and old and old[-1] - trace[-1][1] <= rel_tol * abs(trace[-1][1]):
                        info = 0
                        stalled = True
                        break
                    if it >= maxit:
                        info = 1
                        break
                    nrun = min(self.check_every, maxit - it)
                    t_window = time.perf_counter()
                    if nrun == self.check_every:
                        run_window()                            # the whole window is one graph
                    else:
                        for _ in range(nrun - 1):
                            run_iter()
                        run_iter_checked()                      # last one also measures gpmax (sets DONE when <= eps)
                    hs = st.cpu().numpy()
                    self.window_seconds.append(time.perf_counter() - t_window)      # host-visible time of the window (diagnostics)
                    self._check_replicas(hs)
                    if hs[FAIL] != 0.0:                          # step length underflow or evaluation budget spent (decided on the GPU)
                        info = 2
                        break
                    it, count = int(hs[IT]), int(hs[COUNT])
                    gpmax = float(hs[GPSTATS + 1])
                    trace.append((it, float(hs[F])))
                    checks.append((it, count, float(hs[F])))
                    # line search in trouble: >= 8 trial points per iteration over at least 5 iterations (not just the first
                    # iteration after a restart, whose re-initialised step is expected to backtrack) and nothing gained --
                    # the iterate is stationary to rounding (typical for a restart from an already converged point)
                    back = [c for c in checks if c[0] <= it - 5]
                    if rel_tol > 0.0 and back:
                        it0, count0, f0 = back[-1]
                        if count - count0 >= 8 * (it - it0) and f0 - checks[-1][2] <= rel_tol * abs(checks[-1][2]):
                            info = 0
                            stalled = True
                            break
                hs = st.cpu().numpy()
            finally:
                check(lib.bluest_plan_set_gate(self.hip._h, None, 0))
        if not np.isfinite(hs[F]) or info == 2:
            # a wait between the workgroups of the single-launch projection timed out (never observed; the kernel then returns
            # NaN and raises a sticky flag in its workspace): say so instead of handing back a NaN allocation
            L = self.L
            if L > 4096:
                off = 2 * L + 4 * max((L + 1023) // 1024, 64)          # csrc/spg.hip ProjWs::tau_off
                if float(self.pws[off + 12]) != 0.0:
                    self.pws[off + 12:off + 13].copy_(self._zero)
                    raise RuntimeError("device SPG: the single-launch simplex projection timed out waiting for a workgroup "
                                       "(BLUEST_PROJ_MULTI_LAUNCH=1 selects the multi-launch path)")
        return {"x": self.x.cpu().numpy(), "f": float(hs[F]) * norm, "gpmax": gpmax, "it": int(hs[IT]), "count": int(hs[COUNT]),
                "solver_info": info, "norm": norm, "stalled": stalled}


class ShardedDeviceSpg(DeviceSpg):
    """The device-resident loop over a group-sharded plan (dist.ShardedPlan), one process per GPU: every rank runs the SAME
    iteration on replicated vectors (x, g, d, state) and evaluates its shard of the groups.  One step =

        direction (replicated projection + first trial point)
        Phi records of the shard -> all-reduce(SUM) of the records (peer-write exchange or RCCL)
        -> redundant solve + gradient of the shard + decision (one launch)
        -> gradient scattered into the K_tot-vector -> all-reduce(SUM) (RCCL) -> update

    enqueued on the stream with no host synchronisation in between; the records and the gradient come out of the all-reduces
    bit-identical on every rank, so the replicated state never diverges and all ranks take the same host decisions."""

    def __init__(self, sharded, *args, **kwargs):
        super().__init__(sharded, *args, **kwargs)
        d = dict(dtype=torch.float64, device=self.dev)
        self.rec = torch.empty((1, self.n_out, self.hip.reclen), **d)
        self.gnew = torch.empty((1, self.L), **d)

    def _window_direct(self, n_iterations, check_last):
        import torch.distributed as dist
        lib, h, sh, st = self.lib, self.hip._h, self.plan, self.st
        coef = st.data_ptr() + 8 * COEF                           # dF/dV_o of the accepted trial, written by the decision
        for _ in range(int(n_iterations)):
            self._direction()
            check(lib.bluest_plan_phi(h, self.m.data_ptr(), 1, self.L, self.rec.data_ptr(), _stream()))
            sh.reduce_records(self.rec)
            # redundant solve + gradient tiles of the shard + the line-search decision: one launch
            check(lib.bluest_plan_solve_grad(h, self.rec.data_ptr(), 0.0, self.var.data_ptr(), self.grad.data_ptr(), self.status.data_ptr(),
                                             st.data_ptr(), 1, self.enable.data_ptr(), _stream()))
            check(lib.bluest_plan_combine_grad(h, self.grad.data_ptr(), self.grad.stride(0), coef, self.scale.data_ptr(), 1,
                                               self.gnew.data_ptr(), self.gnew.stride(0), _stream()))
            if sh.world > 1:
                dist.all_reduce(self.gnew, op=dist.ReduceOp.SUM, group=sh.group)
            check(lib.bluest_spg_update(self.x.data_ptr(), self.g.data_ptr(), self.xnew.data_ptr(), self.gnew.data_ptr(), st.data_ptr(),
                                        self.floor, self.L, self.work.data_ptr(), _stream()))
        if check_last:
            self._converged()

    def _check_replicas(self, hs):
        """every rank takes its host decisions from its own copy of the state; the copies are identical by construction
        (deterministic kernels, bit-identical all-reduces).  Should they ever differ, the ranks would leave the loop at different
        windows and the next collective would hang -- so each window ends with ONE tiny all-reduce that makes every rank see
        the disagreement and raise together instead"""
        import torch.distributed as dist
        sh = self.plan
        if sh.world <= 1:
            return
        v = np.array([hs[IT], hs[COUNT], hs[DONE], hs[FAIL], hs[PENDING], hs[F]], dtype=np.float64)
        v = np.where(np.isfinite(v), v, -1.0)
        t = torch.from_numpy(np.concatenate([v, -v]))
        on_dev = dist.get_backend(sh.group) == "nccl"
        if on_dev:
            t = t.to(self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=sh.group)
        t = t.cpu().numpy()
        if not np.array_equal(t[:len(v)], -t[len(v):]):
            raise RuntimeError("sharded SPG: the replicated solver states of the ranks differ (max %s, min %s)" % (t[:len(v)], -t[len(v):]))

    def run(self, x0, **kwargs):
        kwargs["use_graph"] = False                               # collectives are enqueued directly
        return super().run(x0, **kwargs)
