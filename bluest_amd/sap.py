"""
SAP -- single-output sample-allocation problem on the GPU.  Mirrors bluest/sap.py:52-220 for the hot path:
same constructor, same attributes (psi, groups, sizes, cumsizes, invcovs, L, N, e, ES, costs), same closures
variance / variance_GH / get_phi, same solve() signature -- with solver="spg" added (the four third-party
back-ends of the reference are outside this build).
"""
import numpy as np
import torch

from . import misc
from .plan import EVAL_INF, EVAL_NO_MODEL0, EVAL_OK, EVAL_SINGULAR, Plan, simplex_project
from . import host
from .host import host_section, in_host_section
from .spg import spg
from .spg_device import DeviceSpg, ShardedDeviceSpg
from .colgen import colgen_solve

host.warm()     # first-touch work of host_section happens at import, never inside a timed constructor

spg_sap_default_params = {
    "eps": 1.0e-7,            # stop when ||P(x-g)-x||_inf <= eps (objective normalised by its initial value)
    "maxit": 4000,            # total iteration budget over all stages and restarts
    "max_fevals": 10 ** 6,
    "lmbda_min": 10. ** -30,
    "lmbda_max": 10. ** 3,    # cap of the spectral step (objective normalised to 1, x on the unit simplex).  The reference's
                              # 1e30 (meant for its covariance projection) makes every s.y <= 0 event -- frequent on this flat
                              # optimum -- a jump to a vertex followed by ~13 backtracking evaluations; 1e3 was the best of
                              # {1e30, 1e6, 1e3, 1e1} in both speed and objective at K_tot = 245505
    "linesearch_history_length": 10,
    "smoothing_p": (32.0, 2048.0, float("inf")),
                              # multi-output: max_o V_o is replaced by the p-norm (smooth); a tuple = continuation, each stage
                              # warm-started from the previous one; inf = the plain max (mosap.py:145,578), which is what the
                              # last stage polishes.  Of {(32,512), (32,inf), (32,512,inf), (32,2048,inf), (64,2048)} this
                              # schedule gave the best or within 1e-4 of the best max_o V_o on 5 of 5 test problems
                              # (tools/p_matrix.py; the old (32,512) was up to 1.3e-3 worse); inf alone thrashes at the kinks
    "device_loop": True,      # True: whole iteration on the GPU (spg_device.DeviceSpg); False: host-driven bluest_amd.spg.spg
                              # (the host-driven path uses the first smoothing exponent only)
    "slots": 1,               # line-search trial points launched per step of the device loop (a rejected last one carries over)
    "check_every": 20,        # steps between host looks at the device state
    "scaling_floor": 1.0e-8,  # > 0: scaled SPG, steps and projections in the metric diag(1/max(x, floor)); 0: plain SPG
    "method": "newton",       # "newton": multiplicative phase + column generation with a Newton master on the support and a
                              #   certified duality gap (bluest_amd/colgen.py, csrc/newton.hip); "spg": first-order only (also
                              #   the fall-back when the master problem does not fit one workgroup, and for sharded plans)
    "newton": None,           # overrides for colgen_solve (ma_iterations, background, enter_per_round, ...)
    "sparsify_tol": 1.0e-5,   # final support selection: keep the fewest largest entries whose objective is within this (relative)
                              # of the full iterate's (0 = off)
    "prune_tol": 1.0e-7,      # (when that is off) drop the smallest entries holding less than this share of the budget
    "prune_rel": 1.0e-5,      # between restarts: drop entries below prune_rel*max(x) if V does not grow by more than 1e-6
    "restarts": 6,            # restarts of the LAST continuation stage (each re-initialises the spectral step from the pruned point)
    "restart_tol": 1.0e-5,    # stop restarting when a restart improved the objective by less than this (relative)
    "rel_tol": 2.0e-6,        # device loop: also stop when f decreased by less than rel_tol*f over the last
    "stall_window": 60,       #              stall_window iterations (the flat optimum keeps the projected gradient ~1e-4)
}


class BLUESTError(RuntimeError):
    pass


def normalise_groups(groups, K, flatten=True):
    """list over k of int64 arrays (L_k, k); mutates the list in place like sap.py:77 does"""
    flattened = []
    for k in range(1, K + 1):
        gk = groups[k - 1]
        if flatten:
            flattened += [list(g) for g in gk] if not isinstance(gk, np.ndarray) else gk.tolist()
        groups[k - 1] = np.array(gk, dtype=np.int64).reshape((-1, k))
    return flattened


class LazyIndicators(object):
    """the list ES of the reference (sap.py:89-95): ES[i][j] = 1 if model i is in group j.  Only ES[0] (= `e`, the model-0
    constraint) is needed on the path; the N x L table (49 MB at K_tot = 245505) is built when somebody asks for another row."""

    def __init__(self, groups, N, dtype=np.int64):
        self._groups, self._N, self._dtype, self._all, self._e = groups, N, dtype, None, None

    @property
    def e(self):
        """row 0, built on first use (a MOSAP creates one of these per output and normally reads only its own)"""
        if self._e is None:
            L = sum(len(g) for g in self._groups)
            e = np.zeros(L, dtype=self._dtype)
            off = 0
            for gk in self._groups:
                gk = np.asarray(gk)
                if len(gk):
                    e[off:off + len(gk)] = (gk == 0).any(axis=1)
                off += len(gk)
            self._e = e
        return self._e

    def _table(self):
        if self._all is None:
            self._all = [row.astype(self._dtype) for row in indicator_vectors(self._groups, self._N)]
        return self._all

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)) and i == 0:
            return self.e
        return self._table()[i]

    def __len__(self): return self._N
    def __iter__(self): return iter(self._table())


def indicator_vectors(groups, N):
    """ES[i][j] = 1 if model i is in group j (sap.py:89-95), vectorised"""
    L = sum(len(g) for g in groups)
    ES = np.zeros((N, L), dtype=np.int64)
    off = 0
    for gk in groups:
        if len(gk):
            cols = np.repeat(np.arange(off, off + len(gk)), gk.shape[1])
            ES[gk.ravel(), cols] = 1
        off += len(gk)
    return [ES[i] for i in range(N)]


def status_to_python(status, where):
    """reference behaviour for the per-evaluation status codes"""
    if status == EVAL_NO_MODEL0:
        raise AssertionError("%s: model 0 is not sampled (bluest/misc.py:470)" % where)
    if status == EVAL_SINGULAR:
        raise AssertionError("%s: information matrix is singular on the sampled models (bluest/misc.py:473-474)" % where)


def enforce_sample_caps(plan, costs, es, rhs, samples, budget, eps, solver_params, owner, cap_models=None):
    """max_model_samples (bluest/sap.py:222-240): if the unconstrained optimum already respects the caps it is the answer;
    otherwise solve again over the capped set (bluest_amd/capped.py), started from the unconstrained allocation"""
    if len(es) == 0 or samples is None:
        return samples
    if all(float(np.asarray(ee, dtype=np.float64) @ samples) <= rr for ee, rr in zip(es, rhs)):
        return samples
    from .capped import solve_capped
    prm = dict(spg_sap_default_params)
    if solver_params:
        prm.update(solver_params)
    clipped = samples.copy()
    for ee, rr in zip(es, rhs):
        tot = float(np.asarray(ee, dtype=np.float64) @ clipped)
        if tot > rr:
            clipped[np.asarray(ee) > 0] *= rr / tot
    budget_solver = None
    if prm["method"] == "newton" and cap_models is not None and type(getattr(plan, "plan", plan)).__name__ == "Plan":
        cap = {"models": np.asarray(cap_models, dtype=np.int32), "rows": np.stack([np.asarray(ee) for ee in es]), "rhs": np.asarray(rhs, dtype=np.float64)}
        w_h = np.asarray(costs, dtype=np.float64)
        from .capped import cost_shift_capped

        def budget_solver(B, s_norm, start):
            """the second-order finish under shifted costs (capped.cost_shift_capped: the caps never enter the master problem);
            if that fails, the caps as rows of the master's KKT system (colgen.colgen_solve(caps=...), single GPU, <= 64 caps);
            None -> first-order fall-back"""
            x_start = None
            if start is not None:
                x_start = np.maximum(np.asarray(start, dtype=np.float64), 0.0) * w_h
                x_start = x_start / x_start.sum() if x_start.sum() > 0 else None
            m_cs, cinfo = cost_shift_capped(plan, w_h, s_norm, float(B), cap["rows"], cap["rhs"], prm=prm.get("newton"), x_start=x_start)
            candidate = cinfo.get("candidate") if (m_cs is None and isinstance(cinfo, dict)) else None
            xn = None
            if m_cs is None and type(plan).__name__ == "Plan" and len(es) <= 64:
                xn, ninfo = colgen_solve(plan, w_h, s_norm, float(B), prm=prm.get("newton"), caps=cap)
            if m_cs is None and xn is None and candidate is not None:
                m_cs, cinfo = candidate                      # certified to 1e-3 .. 1e-4: better than the uncertified first-order loop
            if m_cs is not None:
                return m_cs, {"it": cinfo["newton_it"], "count": cinfo["full_evals"], "gpmax": cinfo["kkt"], "f": cinfo["F"], "solver_info": 0,
                              "fevals": cinfo["full_evals"], "gevals": cinfo["full_evals"], "pruned": int(len(m_cs) - cinfo["support"]),
                              "method": "newton", "caps": "cost shift (%d free solves)" % cinfo["solves"], "certified_gap": cinfo["gap"],
                              "rounds": cinfo["rounds"], "cap_usage": cinfo["cap_usage"], "multipliers": cinfo["mu"]}
            if xn is None:
                return None
            return (float(B) / w_h) * xn, {"it": ninfo["newton_it"], "count": ninfo["full_evals"] + ninfo["master_evals"], "gpmax": ninfo["kkt"],
                                           "f": ninfo["F"], "solver_info": 0, "fevals": ninfo["full_evals"], "gevals": ninfo["full_evals"],
                                           "pruned": int(len(xn) - ninfo["support"]), "method": "newton", "caps": "rows of the master's KKT system",
                                           "certified_gap": ninfo["gap"],
                                           "rounds": ninfo["rounds"], "cap_usage": ninfo["cap_usage"], "multipliers": ninfo["mu"]}
    m, info = solve_capped(plan, costs, es, rhs, budget=budget, eps=eps, x0=clipped, prm=prm,
                           unconstrained_cost=float(np.asarray(costs) @ samples), budget_solver=budget_solver)
    if m is None:
        raise BLUESTError("SPG with max_model_samples: %s" % info.get("reason", "no feasible allocation"))
    owner.solver_info = info
    return m


def restrict_plan(plan, keep):
    """plan.restrict(keep) with the library's "an output would lose model 0" turned into BLUESTError (what the solver expects)"""
    from ._lib import BluestHipError
    try:
        return plan.restrict(keep)
    except BluestHipError as err:
        if "would not sample model 0" in str(err):
            raise BLUESTError(str(err))
        raise


class SpgAllocator(object):
    """SPG in the scaled variable x = cost*m/B over the unit simplex; all vectors live in HBM.

    objective F(m) = || (V_o(m)/s_o)_o ||_p  (p = inf: max, single output: V itself), minimised over
    {m >= 0, cost.m = B}.  V is homogeneous of degree -1 in m, so the eps-constrained problem
    (min cost s.t. V_o <= eps_o^2) is the B = 1 problem with s_o = eps_o^2 followed by a rescale.
    """

    def __init__(self, plan, costs, e_list, verbose=False, subplan=None):
        self.subplan = subplan     # callable(sorted global group indices) -> Plan restricted to those groups, or None
        self.plan = plan
        self.dev = plan.device
        self.costs = np.asarray(costs, dtype=np.float64)
        self.w = torch.from_numpy(self.costs).to(self.dev)
        self.e_list = e_list  # per output: indicator (global numbering) of groups containing model 0
        self.verbose = verbose

    def _solve_device(self, budget, eps, x0, prm):
        """solve() with the device-resident loop.  All vector bookkeeping between the runs (pruning, support selection, pricing)
        is numpy on the host: the vectors are <= a few MB, and no torch compute operator is touched -- on ROCm the first use
        of each one loads its kernels (30-150 ms a piece), which used to triple the first solve of a process."""
        plan, dev = self.plan, self.dev
        n_out, L, N = plan.n_out, plan.L, plan.N
        s = np.ones(n_out) if budget is not None else np.asarray(eps, dtype=np.float64) ** 2
        w = self.costs

        def to_dev(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)

        def ratios(pl, m_h):
            """max_o V_o/s_o of allocation m_h on plan pl (inf if not evaluable)"""
            var, _, status = pl.eval(m_h, want_grad=False)
            if not (status[0].cpu().numpy() == EVAL_OK).all():
                return np.inf
            r = var[0].cpu().numpy() / s
            return float(r.max()) if np.isfinite(r).all() else np.inf

        def rescale_to_tolerance(m_h):
            """eps mode: scale the allocation so that max_o V_o/eps_o^2 = 1 (V is homogeneous of degree -1).  The scaling can carry an
            entry across the reference's absolute threshold |m| > 1e-6 (bluest/misc.py:453-457: which models count as sampled), which
            moves V by ~1e-8 on ill-conditioned data: repeated until the ratio is 1 to rounding (one pass almost always).  On the
            Navier-Stokes covariances (cond 1.5e11) the evaluation itself moves by 2e-9 when m is scaled by 1 + 1e-10, so the allocation
            returned is the one whose EVALUATED ratio was closest to 1, not the last rescaled one"""
            best = None
            for _ in range(6):
                r_now = ratios(plan, m_h)
                if not np.isfinite(r_now):
                    break
                if best is None or abs(r_now - 1.0) < best[0]:
                    best = (abs(r_now - 1.0), m_h)
                if abs(r_now - 1.0) <= 1.0e-12:
                    break
                m_h = m_h * r_now
            return m_h if best is None else best[1]

        if budget is not None:
            B = float(budget)
        else:
            # eps mode: pick the working budget so that the uniform start already has max_o V_o/eps_o^2 ~ 1; sample counts
            # then have their real magnitude for the reference's absolute thresholds (|m| > 1e-6, max|m| >= 0.05)
            B_try = 1.0e6 * float(w.max())
            r_try = ratios(plan, np.full(L, 1.0 / L) * (B_try / w))
            if not np.isfinite(r_try):
                raise BLUESTError("SPG: the uniform allocation is infeasible (model 0 unsampled or singular information matrix)")
            B = B_try * r_try
        p_list = prm["smoothing_p"] if isinstance(prm["smoothing_p"], (list, tuple)) else [prm["smoothing_p"]]
        p_list = [float(q) for q in p_list] if n_out > 1 else [np.inf]
        scale_h = B / w                                            # m = scale * x
        scale = None                                               # (device copy: the first-order loop makes it when it runs)
        floor = float(prm["scaling_floor"])
        x = np.full(L, 1.0 / L) if x0 is None else np.asarray(x0, dtype=np.float64) * w / B
        if x0 is not None:                                         # (the uniform start is on the simplex already)
            x = simplex_project(to_dev(x), want_d=False)[0].cpu().numpy()
        if not np.isfinite(ratios(plan, scale_h * x)):
            raise BLUESTError("SPG: the initial allocation does not sample model 0 / is infeasible")
        tot = {"it": 0, "count": 0}
        if prm["method"] == "newton" and type(getattr(plan, "plan", plan)).__name__ == "Plan":      # a HIP plan, or a ShardedPlan over HIP plans
            import os
            trace = print if os.environ.get("BLUEST_COLGEN_LOG") else None       # debugging aid: the rounds of the second-order finish
            xn, ninfo = colgen_solve(plan, w, s, B, x0=None if x0 is None else x, prm=prm.get("newton"), log=trace)
            if trace and xn is None:
                trace("second-order finish gave up: %s" % (ninfo,))
            if xn is not None:
                m = scale_h * xn
                if budget is None:
                    m = rescale_to_tolerance(m)
                self.info = {"it": ninfo["newton_it"], "count": ninfo["full_evals"] + ninfo["master_evals"], "gpmax": ninfo["kkt"],
                             "f": ninfo["F"], "solver_info": 0, "fevals": ninfo["full_evals"], "gevals": ninfo["full_evals"],
                             "pruned": int(L - ninfo["support"]), "method": "newton", "certified_gap": ninfo["gap"],
                             "rounds": ninfo["rounds"], "master_evals": ninfo["master_evals"], "multipliers": ninfo["mu"]}
                if "host_ms" in ninfo:
                    self.info["host_ms"] = ninfo["host_ms"]
                cert = ninfo.get("certificate")
                if cert is not None:
                    # the certificate in the caller's terms: an allocation of cost B (the working budget), multipliers, and the
                    # weight of the uniform background at which the bound  F*_B >= lower_bound  was obtained
                    mc = np.zeros(L)
                    mc[cert["support"]] = scale_h[cert["support"]] * np.maximum(cert["x"], 0.0) / max(float(np.maximum(cert["x"], 0.0).sum()), 1e-300)
                    self.info["certificate"] = {"allocation": mc, "multipliers": cert["mu"], "background": cert["background"],
                                                "lower_bound": cert["lower_bound"], "budget": B, "scales": s.copy()}
                return m
            ex = getattr(plan, "exchange", None)
            if ex is not None and ex.timed_out():
                # (collective: a time-out poisons the record with NaN on the ranks that saw it; every rank ends here because the
                # non-finite iterate is detected on gathered data)
                raise BLUESTError("sharded solve: the peer-write exchange of the Phi records timed out (a rank is missing or stalled)")
            if self.verbose:
                print("second-order finish unavailable (%s); first-order SPG only" % ninfo)

        def run_stages(pl, sc_h, sc, xc, stages, polish_last, loose=5.0, window=None, slots=None):
            """the continuation stages on plan `pl` (variables scaled by sc) from xc; polish_last: the last stage gets the
            restarts and the full stall tolerance.  Returns the last run's result (x as numpy), or None without budget left"""
            def prune_dust(xq):
                rel = float(prm["prune_rel"])
                if rel <= 0.0:
                    return xq
                xp = np.where(xq < rel * xq.max(), 0.0, xq)
                xp = xp / xp.sum()
                return xp if ratios(pl, sc_h * xp) <= ratios(pl, sc_h * xq) * (1.0 + 1.0e-6) else xq

            # ONE solver object for all stages: the smoothing exponent lives in the device state, so the captured hipGraphs
            # are shared by the stages
            loop_cls = ShardedDeviceSpg if hasattr(pl, "reduce_records") else DeviceSpg     # dist.ShardedPlan: collective loop
            dspg = loop_cls(pl, sc, s, stages[0], floor, lmbda_min=prm["lmbda_min"], lmbda_max=prm["lmbda_max"],
                             Hlength=prm["linesearch_history_length"], slots=prm["slots"] if slots is None else slots,
                             check_every=prm["check_every"])
            out = None
            for stage, pq in enumerate(stages):
                dspg.p = float(pq)
                f_prev = None
                last_stage = polish_last and stage == len(stages) - 1
                # the earlier stages minimise a surrogate (a looser smooth max): one run to a looser stall tolerance is all the
                # warm start needs; the restarts and the full tolerance are spent on the last stage only
                maxit_left = int(prm["maxit"]) - tot["it"] if last_stage else int(prm["maxit"]) // len(p_list)
                for restart in range((int(prm["restarts"]) if last_stage else 0) + 1):
                    if maxit_left <= 0:
                        break
                    out = dspg.run(xc, eps=prm["eps"], maxit=maxit_left, max_fevals=prm["max_fevals"],
                                   rel_tol=prm["rel_tol"] * (1.0 if last_stage else loose),
                                   stall_window=prm["stall_window"] if window is None else window)
                    xc = prune_dust(out["x"])
                    out["x"] = xc
                    tot["it"] += out["it"]
                    tot["count"] += out["count"]
                    maxit_left -= out["it"]
                    f_abs = out["f"]
                    if out["solver_info"] == 0 and not out["stalled"]:
                        break                                          # converged in the projected-gradient sense
                    if f_prev is not None and f_prev - f_abs <= float(prm["restart_tol"]) * abs(f_abs):
                        break
                    f_prev = f_abs
            return out

        if scale is None:
            scale = to_dev(scale_h)
        res = run_stages(plan, scale_h, scale, x, p_list, True)
        x = res["x"]
        xs = x
        # support selection by objective: a first-order iterate keeps hundreds of entries that together hold ~1e-4 of the
        # budget (the optimum sits on <= N entries per output; the reference's SDP solvers return such a point).  Keep the S
        # entries with the largest cost share, give the rest of the budget to them, and take the smallest S (doubling from N)
        # whose objective is within sparsify_tol of the full iterate's: a handful of 15 us evaluations.  Without this the integer
        # projection faces hundreds of fractional entries below one sample and cannot find a feasible point.
        pruned = 0
        stol, tol = float(prm["sparsify_tol"]), float(prm["prune_tol"])
        if stol > 0.0:
            order = np.argsort(-xs, kind="stable")
            nnz = int((xs > 0).sum())
            f0 = ratios(plan, scale_h * xs)
            S = N
            while S < nnz:
                xp = np.zeros(L)
                xp[order[:S]] = xs[order[:S]]
                xp = xp / xp.sum()
                if ratios(plan, scale_h * xp) <= f0 * (1.0 + stol):
                    xs, pruned = xp, L - S
                    break
                S *= 2
        if tol > 0.0 and pruned == 0:
            # drop the smallest entries whose cumulative cost share is below prune_tol and give their budget to the rest
            order = np.argsort(xs, kind="stable")
            k = int((np.cumsum(xs[order]) <= tol).sum())
            if 0 < k < L:
                xp = xs.copy()
                xp[order[:k]] = 0.0
                xp = xp / xp.sum()
                if ratios(plan, scale_h * xp) <= ratios(plan, scale_h * xs) * (1.0 + 10.0 * tol):
                    xs, pruned = xp, k
        m = scale_h * xs
        if budget is None:
            m = rescale_to_tolerance(m)
        self.info = {"it": tot["it"], "count": tot["count"], "gpmax": res["gpmax"], "f": res["f"], "solver_info": res["solver_info"],
                     "fevals": tot["count"], "gevals": tot["it"] + len(p_list), "pruned": pruned}
        return m

    def solve(self, budget=None, eps=None, x0=None, params=None):
        prm = dict(spg_sap_default_params)
        if params:
            prm.update(params)
        if prm["device_loop"]:
            return self._solve_device(budget, eps, x0, prm)
        plan = self.plan
        n_out = plan.n_out
        s = np.ones(n_out) if budget is not None else np.asarray(eps, dtype=np.float64) ** 2
        if budget is not None:
            B = float(budget)
        else:
            # eps mode: pick the working budget so that the uniform start already has max_o V_o/eps_o^2 ~ 1; sample counts
            # then have their real magnitude for the reference's absolute thresholds (|m| > 1e-6, max|m| >= 0.05)
            B_try = 1.0e6 * float(self.costs.max())
            m_try = torch.full((plan.L,), 1.0 / plan.L, dtype=torch.float64, device=self.dev) * (B_try / self.w)
            v_try, _, st_try = plan.eval(m_try, want_grad=False)
            r_try = (v_try[0].cpu().numpy() / s)
            if not (st_try[0].cpu().numpy() == EVAL_OK).all() or not np.isfinite(r_try).all():
                raise BLUESTError("SPG: the uniform allocation is infeasible (model 0 unsampled or singular information matrix)")
            B = B_try * float(r_try.max())
        p_list = prm["smoothing_p"] if isinstance(prm["smoothing_p"], (list, tuple)) else [prm["smoothing_p"]]
        p_list = [float(q) for q in p_list] if n_out > 1 else [np.inf]
        p = p_list[0]
        scale = B / self.w                                         # m = scale * x
        st = {"x": None, "var": None, "status": None, "fevals": 0, "gevals": 0, "norm": 1.0}

        def objective(var):
            r = var / s
            if not np.isfinite(r).all():
                return np.inf, None
            rmax = r.max()
            if np.isinf(p):
                coef = np.zeros(n_out)
                coef[int(np.argmax(r))] = 1.0
                return rmax, coef / s
            t = (r / rmax) ** p
            F = rmax * t.sum() ** (1.0 / p)
            coef = (r / rmax) ** (p - 1) * t.sum() ** (1.0 / p - 1.0)   # dF/dr_o
            return F, coef / s

        def evaluate(x, want_grad):
            m = scale * x
            var, grad, status = plan.eval(m, want_grad=want_grad)
            var_h = var[0].cpu().numpy()
            status_h = status[0].cpu().numpy()
            ok = (status_h == EVAL_OK).all()
            F, coef = objective(var_h) if ok else (np.inf, None)
            return F, coef, grad, m

        def feval(x):
            st["fevals"] += 1
            F, coef, _, _ = evaluate(x, False)
            return F / st["norm"]

        def geval(x):
            st["gevals"] += 1
            F, coef, grad, _ = evaluate(x, True)
            if coef is None:
                raise BLUESTError("SPG: gradient requested at an infeasible point")
            c = torch.from_numpy(coef / st["norm"]).to(self.dev).reshape(1, -1)
            return plan.combine_grad(grad, c, scale=scale)[0]

        floor = float(prm["scaling_floor"])

        def proj(x):
            return simplex_project(x, want_d=False)[0]

        def proj_step(x, g, lmbda):
            _, d, stats = simplex_project(x, g, lmbda, want_p=False, floor=floor)
            sh = stats.cpu().numpy()
            return d, float(sh[0]), float(sh[1])

        def metric_dot(s_, x_):
            return float((s_ * s_ / torch.clamp(x_, min=floor)).sum())

        L = plan.L
        if x0 is None:
            x = torch.full((L,), 1.0 / L, dtype=torch.float64, device=self.dev)
        else:
            x = torch.from_numpy(np.asarray(x0, dtype=np.float64)).to(self.dev) * self.w / B
        x = proj(x)
        F0 = feval(x)
        if not np.isfinite(F0):
            raise BLUESTError("SPG: the initial allocation does not sample model 0 / is infeasible")
        st["norm"] = F0

        hist = []

        def stall(it, f, gpmax, lmbda):
            hist.append(f)

        # host-driven loop (the reference's driver with GPU callbacks; device_loop=True goes through _solve_device)
        res = spg(feval, geval, proj, x, eps=prm["eps"], maxit=prm["maxit"], max_fevals=prm["max_fevals"], verbose=self.verbose,
                  lmbda_min=prm["lmbda_min"], lmbda_max=prm["lmbda_max"], Hlength=prm["linesearch_history_length"],
                  proj_step=proj_step, callback=stall, metric_dot=metric_dot if floor > 0 else None)
        xs = res["x"]
        # prune the dust: SPG iterates keep thousands of entries with a negligible share of the budget (the reference's SDP
        # returns a sparse point); drop the smallest entries whose cumulative cost share is below prune_tol and give their
        # budget to the rest -- changes the objective by O(prune_tol), checked below
        tol = float(prm["prune_tol"])
        pruned = 0
        stol = float(prm["sparsify_tol"])
        if stol > 0.0:
            # support selection by objective: a first-order iterate keeps hundreds of entries that together hold ~1e-4 of the
            # budget (the optimum sits on <= N entries per output; the reference's SDP solvers return such a point).  Keep the
            # S entries with the largest cost share, give the rest of the budget to them, and take the smallest S (doubling from
            # N) whose objective is within sparsify_tol of the full iterate's: a handful of 15 us evaluations.  Without this the
            # integer projection faces hundreds of fractional entries below one sample and cannot find a feasible point.
            order = torch.argsort(xs, descending=True)
            nnz = int((xs > 0).sum())
            v0, _, s0 = plan.eval(scale * xs, want_grad=False)
            f0 = float((v0[0] / torch.from_numpy(s).to(self.dev)).max())
            S = plan.N
            while S < nnz:
                xp = torch.zeros_like(xs)
                xp[order[:S]] = xs[order[:S]]
                xp = xp / xp.sum()
                vp, _, sp = plan.eval(scale * xp, want_grad=False)
                if bool((sp == EVAL_OK).all()) and float((vp[0] / torch.from_numpy(s).to(self.dev)).max()) <= f0 * (1.0 + stol):
                    xs, pruned = xp, L - S
                    break
                S *= 2
        if tol > 0.0 and pruned == 0:
            xs_sorted, order = torch.sort(xs)
            cum = torch.cumsum(xs_sorted, 0)
            k = int((cum <= tol).sum())
            if 0 < k < L:
                xp = xs.clone()
                xp[order[:k]] = 0.0
                xp = xp / xp.sum()
                vp, _, sp = plan.eval(scale * xp, want_grad=False)
                v0, _, _ = plan.eval(scale * xs, want_grad=False)
                ok = bool((sp == EVAL_OK).all()) and float(((vp[0] / v0[0]).max())) <= 1.0 + 10.0 * tol
                if ok:
                    xs, pruned = xp, k
        m = (scale * xs)
        if budget is None:
            # rescale so that max_o V_o/eps_o^2 = 1 (V homogeneous of degree -1)
            var, _, status = plan.eval(m, want_grad=False)
            r = (var[0].cpu().numpy() / s).max()
            m = m * r
        self.info = {"it": res["it"], "count": res["count"], "gpmax": res["gpmax"], "f": res["f"] * st["norm"],
                     "solver_info": res["solver_info"], "fevals": st["fevals"], "gevals": st["gevals"], "pruned": pruned}
        return m.cpu().numpy()


def estimator_rhs(N, K, cumsizes, groups, invcovs, sums, m):
    """y = sum_i R_i^T C_i^-1 sums_i over the SAMPLED groups (bluest/sap.py:104-110; the reference walks all L groups in Python:
    minutes at K_tot = 245505, the unsampled ones contribute invcov * 0).  Per group size the products C_i^-1 sums_i are one einsum
    and the scatter into y one np.add.at; sums that are not plain numbers (user objects with + and *) take the generic accumulation.
    Returns (y as an (N, ...) array or None, {model: accumulated object})."""
    y_num, y_obj = None, {}
    for k in range(1, K + 1):
        hit = np.flatnonzero(m[cumsizes[k - 1]:cumsizes[k]] != 0)
        if len(hit) == 0:
            continue
        members = np.asarray(groups[k - 1])[hit]                                       # (h, k) model indices
        blocks = np.asarray(invcovs[k - 1]).reshape(-1, k, k)[hit]                    # (h, k, k)
        mine = [sums[cumsizes[k - 1] + i] for i in hit.tolist()]
        try:
            vals = np.asarray(mine, dtype=np.float64)                                  # (h, k) or (h, k, ...) for array-valued outputs
            numeric = vals.ndim >= 2 and vals.shape[:2] == members.shape
        except (TypeError, ValueError):
            numeric = False
        if numeric:
            contrib = np.einsum("hjs,hs...->hj...", blocks, vals)
            if y_num is None:
                y_num = np.zeros((N,) + contrib.shape[2:])
            np.add.at(y_num, members, contrib)
        else:
            for g, blk, sm in zip(members.tolist(), blocks, mine):
                for j, model in enumerate(g):
                    for c, term in zip(blk[j].tolist(), sm):
                        y_obj[model] = y_obj[model] + c * term if model in y_obj else c * term
    return y_num, y_obj


def sample_cap_rows(ES, N, max_model_samples):
    """bluest/sap.py:222-240, bluest/mosap.py:326-344: (indicator rows ES[i], integer caps) of the models with a finite cap; the
    argument is checked as the reference checks it (numpy array, one entry per model, model 0 may be sampled at least once)"""
    if max_model_samples is None:
        return [], []
    caps = max_model_samples
    if not isinstance(caps, np.ndarray) or len(caps) != N:
        raise ValueError("max_model_samples must be a numpy array with one entry per model (%d); np.inf leaves a model uncapped" % N)
    if caps[0] < 1:
        raise ValueError("max_model_samples[0] < 1: the high-fidelity model needs at least one sample")
    capped = np.flatnonzero(np.isfinite(caps))
    return [ES[i] for i in capped], [int(np.round(caps[i])) for i in capped]


class SAP(object):
    def __init__(self, C, K, groups, costs, verbose=True, device=None, max_candidates=1):
        """bluest/sap.py:53-97.  `groups` (list over k of lists of model tuples) is converted in place to int64
        arrays, as the reference does at :77.  The per-group pseudo-inverses (:69-79) are computed on the GPU."""
        self.verbose = verbose
        self.C = C
        self.N = C.shape[0]
        self.K = K
        self.costs = costs
        self.samples = None
        self.budget = None
        self.eps = None
        self.tot_cost = None
        with host_section():
            self._build(C, K, groups, device, max_candidates)

    def _build(self, C, K, groups, device, max_candidates):
        sizes = [0] + [len(groupsk) for groupsk in groups]
        self.flattened_groups = normalise_groups(groups, K)
        self.sizes = sizes
        self.groups = groups
        self.cumsizes = np.cumsum(sizes)
        self.L = int(self.cumsizes[-1])

        self.plan = Plan(self.N, self.L, [{"K": K, "sizes": sizes[1:], "groups": groups, "C": np.asarray(C, dtype=np.float64),
                                           "mapping": None}], max_candidates=max_candidates, device=device)
        self._plan_output = 0              # which output of self.plan holds this SAP's inverses
        self.ES = LazyIndicators(groups, self.N)
        self.e = self.ES[0]
        self._psi = None

    @property
    def invcovs(self):
        """list over k of the flattened pseudo-inverses pinv(C[g, g]) (sap.py:69-79).  They are computed and kept on the GPU;
        this host copy is made when somebody asks for it"""
        if self.__dict__.get("_invcovs") is None:
            flat = self._inverse_source()
            out, off = [], 0
            for k in range(1, self.K + 1):
                n = self.sizes[k] * k * k
                out.append(flat[off:off + n] if n > 0 else np.array([]))
                off += n
            self.__dict__["_invcovs"] = out
        return self.__dict__["_invcovs"]

    def _inverse_source(self):
        return self.plan.invcovs[self._plan_output]

    def _gather_inverses(self, local_idx):
        """inverses of some of this SAP's groups (local indices) without bringing all of them to the host"""
        if self.__dict__.get("_invcovs") is not None:
            ks = np.searchsorted(self.cumsizes, local_idx, side="right")
            return np.concatenate([self._invcovs[k - 1][(i - self.cumsizes[k - 1]) * k * k:(i - self.cumsizes[k - 1] + 1) * k * k]
                                   for i, k in zip(local_idx, ks)]) if len(local_idx) else np.zeros(0)
        pl, o = self._inverse_plan()
        return pl.gather_invcovs(o, local_idx)

    def _inverse_plan(self):
        return self.plan, self._plan_output

    # ---- psi is only needed by SDP solvers / integer projection: assembled on demand (sap.py:129) --------
    @property
    def psi(self):
        if self._psi is None:
            self._psi = np.hstack([misc.assemble_psi(self.N, k, self.sizes[k], self.groups[k - 1], self.invcovs[k - 1])
                                   for k in range(1, self.K + 1) if len(self.groups[k - 1]) > 0])
        return self._psi

    def get_variance_functions(self):
        """bluest/sap.py:121-143 creates the closures get_phi / variance / variance_GH here; in this class they are ordinary
        methods (same names, same call shapes): a closure stored on the instance that refers back to the instance is a reference
        cycle, and then the plan's HBM and the solver's hipGraphs live until the cyclic collector happens to run.  With methods
        everything is released the moment the last reference to the SAP goes away."""
        return self.get_phi, self.variance, self.variance_GH

    def get_phi(self, m, delta=0):
        """bluest/sap.py:131-132, bluest/misc.py:459-461"""
        PHI = self.plan.phi_matrix(m, delta=delta)[0, 0]
        return PHI if isinstance(m, torch.Tensor) else PHI.cpu().numpy()

    def variance(self, m, delta=0):
        """bluest/sap.py:133-134, bluest/misc.py:463-477"""
        var, _, status = self.plan.eval(m, delta=delta, want_grad=False)
        st = int(status[0, 0])
        if st == EVAL_INF:
            return np.inf
        status_to_python(st, "variance")
        return float(var[0, 0])

    def variance_GH(self, m, delta=0, nohess=False):
        """bluest/sap.py:135-136, bluest/misc.py:479-505"""
        var, grad, status = self.plan.eval(m, delta=delta, want_grad=True)
        st = int(status[0, 0])
        g = grad[0] if isinstance(m, torch.Tensor) else grad[0].cpu().numpy()
        if st == EVAL_INF:
            return np.inf, g      # the reference returns a 2-tuple here (misc.py:484)
        if st == EVAL_SINGULAR:
            # rank-deficient restricted Phi: the reference's variance_GH goes through numpy's pinv (misc.py:487,490) and returns a
            # finite value; so does the Jacobi eigen-pinv path of the plan (rare, a second evaluation)
            var, grad, status = self.plan.eval_pinv(m, delta=delta)
            g = grad[0] if isinstance(m, torch.Tensor) else grad[0].cpu().numpy()
        V = float(var[0, 0])
        if nohess:
            return V, g, None
        return V, g, self._hessian(m, delta)

    def _restricted_plan(self, keep):
        """plan of this problem restricted to the groups `keep` (sorted global indices); the stored pseudo-inverses are reused"""
        keep = np.asarray(keep, dtype=np.int64)
        if self.plan.n_out == 1 and self._plan_output == 0:
            return restrict_plan(self.plan, keep)                        # native: bluest_plan_restrict
        groups, invcovs, sizes = [], [], []
        for k in range(1, self.K + 1):
            lo, hi = self.cumsizes[k - 1], self.cumsizes[k]
            sel = keep[(keep >= lo) & (keep < hi)]
            groups.append(np.asarray(self.groups[k - 1]).reshape(-1, k)[sel - lo])
            invcovs.append(self._gather_inverses(sel))                   # k x k blocks of the kept groups only
            sizes.append(len(sel))
        if not any(len(g) and (g == 0).any() for g in groups):
            raise BLUESTError("restricted plan would not sample model 0")
        return Plan(self.N, len(keep), [{"K": self.K, "sizes": sizes, "groups": groups, "invcovs": invcovs, "mapping": None}],
                    max_candidates=1, device=self.plan.device)

    def get_cleanup_matrix(self, m, delta=0):
        """bluest/misc.py:507-516 (`assemble_cleanup_matrix`), bluest/sap.py:137-138: X[:, i] = R_i^T C_i^-1 (pinv(Phi)[0])_g --
        allocations that differ by a null vector of X have the same estimator variance to first order"""
        m = np.asarray(m.cpu().numpy() if isinstance(m, torch.Tensor) else m, dtype=np.float64)
        if abs(m).max() < 0.05:
            raise ValueError("No entry greater or equal than 1 found in m.")
        invPHI = np.linalg.pinv(np.asarray(self.get_phi(m, delta=delta)))
        return np.hstack([misc.cleanupK(k, self.sizes[k], self.groups[k - 1], self.invcovs[k - 1], invPHI)
                          for k in range(1, self.K + 1) if self.sizes[k] > 0])

    def _hessian(self, m, delta):
        """bluest/misc.py:497-503: K x K blocks of hessKQ, then hess += hess.T.  O(L^2) memory by construction."""
        m_h = m.cpu().numpy() if isinstance(m, torch.Tensor) else np.asarray(m, dtype=np.float64)
        rec = self.plan.phi(m_h)
        N = self.N
        PHI = rec[0, 0, :N * N].reshape(N, N) + float(delta) * torch.eye(N, dtype=torch.float64, device=rec.device)
        invPHI = torch.linalg.pinv(PHI).cpu().numpy()
        L, K, cs = self.L, self.K, self.cumsizes
        hess = np.zeros((L, L))
        for k in range(1, K + 1):
            for q in range(1, K + 1):
                if self.sizes[k] and self.sizes[q]:
                    hess[cs[k - 1]:cs[k], cs[q - 1]:cs[q]] = misc.hessKQ(k, q, self.sizes[k], self.sizes[q], self.groups[k - 1],
                                                                          self.groups[q - 1], self.invcovs[k - 1],
                                                                          self.invcovs[q - 1], invPHI)
        hess += hess.T
        return hess

    def compute_BLUE_estimator(self, sums, samples=None):
        """bluest/sap.py:99-119 + bluest/misc.py:518-544 (PHIinvY0): y = sum_i R_i^T C_i^-1 sums_i (estimator_rhs, host), then
        mu = sum_j pinv(PHI[idx])[0,j] y_j with row 0 of the pseudo-inverse from the GPU solve"""
        if samples is None: samples = self.samples
        m = np.asarray(samples, dtype=np.float64)
        y_num, y_obj = estimator_rhs(self.N, self.K, self.cumsizes, self.groups, self.invcovs, sums, m)
        if abs(m).max() < 0.05: return np.inf
        rec = self.plan.phi(m)
        var, v, status = self.plan.solve(rec)
        status_to_python(int(status[0, 0]), "compute_BLUE_estimator")
        v = v[0, 0].cpu().numpy()
        mu = 0
        for j in np.flatnonzero(v).tolist():
            if y_num is not None:
                mu = mu + v[j] * y_num[j]
            if j in y_obj:
                mu = mu + v[j] * y_obj[j]
        return mu, float(var[0, 0])

    def get_max_sample_constraints(self, max_model_samples):
        """bluest/sap.py:222-240"""
        return sample_cap_rows(self.ES, self.N, max_model_samples)

    @in_host_section
    def solve(self, budget=None, eps=None, solver="spg", x0=None, continuous_relaxation=False, max_model_samples=None,
              solver_params=None):
        """bluest/sap.py:189-220 with solver="spg" (the reference's cvxpy/cvxopt/ipopt/scipy back-ends are
        third-party and not part of this build)."""
        if budget is None and eps is None:
            raise ValueError("Need to specify either budget or RMSE tolerance")
        if solver not in ["spg", "scipy", "cvxpy", "ipopt", "cvxopt"]:
            raise ValueError("Optimization solvers available: 'spg' (this build); the reference also lists 'scipy', 'ipopt', 'cvxopt', 'cvxpy'")
        if solver != "spg":
            raise BLUESTError("solver=%r is a third-party back-end of the reference that this GPU build does not ship; use solver='spg'" % solver)
        es, rhs = self.get_max_sample_constraints(max_model_samples)      # validates the argument as the reference does

        if self.verbose:
            if eps is None: print("Minimizing statistical error for fixed cost...\n")
            else:           print("Minimizing cost given statistical error tolerance...\n")

        alloc = SpgAllocator(self.plan, self.costs, [self.e], verbose=False, subplan=self._restricted_plan)
        try:
            samples = alloc.solve(budget=budget, eps=None if eps is None else [eps], x0=x0, params=solver_params)
            self.solver_info = alloc.info
            cap_models = None if max_model_samples is None else [i for i in range(self.N) if np.isfinite(max_model_samples[i])]
            samples = enforce_sample_caps(self.plan, self.costs, es, rhs, samples, budget, None if eps is None else [eps],
                                          solver_params, self, cap_models=cap_models)
        except BLUESTError as err:
            if self.verbose: print(str(err))
            self.samples = None
            return None
        if samples is None:
            self.samples = None
            return None
        if samples @ self.e < 1.0 - 1.0e-9:
            if self.verbose: print("SPG solution samples model 0 less than once; infeasible for this budget.")
            self.samples = None
            return None

        if not continuous_relaxation:
            from .integer import integer_projection_sap
            try:
                samples = integer_projection_sap(self, samples, budget=budget, eps=eps, max_model_samples=max_model_samples)
            except AssertionError as err:
                print(str(err))
                self.samples = None
                return None

        self.samples = samples
        self.budget = budget
        self.eps = eps
        self.tot_cost = samples @ self.costs
        return samples
