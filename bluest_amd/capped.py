"""
solver="spg" with `max_model_samples` (bluest/sap.py:189-240, bluest/mosap.py:291-344): per-model caps  ES[i] . m <= nmax_i  on
top of  cost . m <= budget  (bluest/sap.py:362,407).

In the scaled variable x = cost*m/B the feasible set is  { x >= 0, sum x + slack = 1, a_i . x <= nmax_i }  with
a_i[g] = ES[i][g] * B / cost[g]; the slack entry is the unspent part of the budget (with caps the optimum need not spend all of
it).  SPG needs the projection onto that set in the diagonal metric diag(1/s), s = max(x, floor): its dual has one multiplier
nu_i >= 0 per cap and  p(nu) = P_simplex^s(u - s*A^T nu),  so every dual evaluation is ONE launch of the simplex-projection
kernel (csrc/spg.hip); the multipliers are found by a semismooth Newton iteration (the dual is piecewise quadratic:
Hessian A W A^T with W = diag(s_F) - s_F s_F^T / sum s_F on the positive entries F) with backtracking on the dual value, about
3 kernel launches per projection.  The spectral step is capped at 1 here (`capped_lmbda_max`): with huge steps all mass sits on
one entry, the dual Hessian vanishes and Newton degenerates (measured: cap 1 -> 0 fall-backs and 3 launches per iteration,
cap 10 -> 511 fall-backs and 360 launches per iteration for the same objective).
The loop around it is the host-driven reference driver (bluest_amd.spg.spg); this option is rarely used and not the hot path.
"""
import numpy as np
import torch

from .plan import EVAL_OK, simplex_project
from .spg import spg


DEBUG = False
TRACE_FALLBACK = False


class CappedSimplex(object):
    def __init__(self, A, c, floor):
        """A: (k, L1) torch float64 on the device (rows a_i, zero in the slack column); c: (k,) torch"""
        self.A, self.c, self.floor = A, c, float(floor)
        self.nu = torch.zeros(A.shape[0], dtype=torch.float64, device=A.device)
        self.launches = 0

    def project(self, x, g=None, lmbda=0.0):
        """p = argmin sum (p - u)^2 / s over the capped simplex, u = x - lmbda*s*g (g None: u = x).  Returns (p, d, g.d, max|d|)"""
        A, c, floor = self.A, self.c, self.floor
        lam_g = torch.zeros_like(x) if g is None else lmbda * g
        s = torch.clamp(x, min=floor) if floor > 0 else torch.ones_like(x)
        u = x - s * lam_g
        tol = 1.0e-12
        cmax = max(1.0, float(c.abs().max()))

        def dual(nu_):
            """value of the (concave) dual, the minimiser p(nu) and the dual gradient r = A p - c"""
            p_ = simplex_project(x, lam_g + A.T @ nu_, 1.0, want_d=False, floor=floor)[0]
            self.launches += 1
            r_ = A @ p_ - c
            return float(0.5 * ((p_ - u) ** 2 / s).sum() + nu_ @ r_), p_, r_

        def newton(nu):
            """semismooth Newton on the dual from nu, backtracking on the dual value; returns (residual, p, nu)"""
            D, p, r = dual(nu)
            best = None
            for it in range(40):
                res = float(torch.where(nu > 0, r.abs(), torch.clamp(r, min=0.0)).max()) / cmax
                if DEBUG:
                    print("    capped projection it %d: res %.3e dual %.6e nu %s nF %d" % (it, res, D, nu.cpu().numpy(), int((p > 0).sum())))
                if best is None or res < best[0]:
                    best = (res, p, nu.clone())
                if res <= tol:
                    break
                act = (nu > 0) | (r > 0)
                F = p > 0
                sF = torch.where(F, s, torch.zeros_like(s))
                Aa = A[act]
                As = Aa * sF
                H = As @ Aa.T - torch.outer(As.sum(dim=1), As.sum(dim=1)) / sF.sum()
                H = H + 1.0e-12 * torch.eye(H.shape[0], dtype=H.dtype, device=H.device) * float(H.diagonal().abs().max() + 1e-300)
                try:
                    step = torch.linalg.solve(H, r[act])
                except Exception:
                    break
                if not bool(torch.isfinite(step).all()):
                    break
                # semismooth Newton is only locally convergent (the Hessian changes with the support of p): backtrack on the dual
                moved = False
                t = 1.0
                for _ in range(30):
                    new = nu.clone()
                    new[act] = torch.clamp(nu[act] + t * step, min=0.0)
                    Dn, pn, rn = dual(new)
                    if Dn >= D + 1.0e-4 * float(r @ (new - nu)) and np.isfinite(Dn):
                        moved = not torch.equal(new, nu)
                        nu, D, p, r = new, Dn, pn, rn
                        break
                    t *= 0.5
                if not moved:
                    break
            res_last = float(torch.where(nu > 0, r.abs(), torch.clamp(r, min=0.0)).max()) / cmax
            if best is None or res_last < best[0]:
                best = (res_last, p, nu.clone())
            return best

        # warm start: the multipliers scale with the step length and the active caps rarely change between iterations; an
        # over-sized start leaves no mass on the capped groups (vanishing dual Hessian), so a failed warm run restarts cold
        warm = self.nu * float(lmbda) if g is not None else torch.zeros_like(self.nu)
        best = newton(warm)
        if best[0] > 1.0e-9 and bool((warm > 0).any()):
            cold = newton(torch.zeros_like(self.nu))
            best = cold if cold[0] < best[0] else best
        res, p, nu = best
        if res <= 1.0e-9 and bool(torch.isfinite(p).all()):
            if g is not None and lmbda > 0:
                self.nu = nu / float(lmbda)
        else:
            # degenerate dual (e.g. a huge spectral step puts all mass on one entry: the dual Hessian vanishes and Newton has
            # nothing to work with): fall back to the furthest feasible point on the segment from x (feasible) to the plain
            # simplex projection p0 -- not the projection, but a feasible descent step that keeps the iteration going
            self.fallbacks = getattr(self, "fallbacks", 0) + 1
            if DEBUG or TRACE_FALLBACK:
                print("    capped projection FALLBACK: best res %.3e, lmbda %g, nu %s" % (res, lmbda, nu.cpu().numpy()))
            p0 = simplex_project(x, lam_g, 1.0, want_d=False, floor=floor)[0]
            self.launches += 1
            step = p0 - x
            rate = A @ step
            room = c - A @ x
            t = torch.where(rate > 0, torch.clamp(room, min=0.0) / torch.clamp(rate, min=1e-300), torch.ones_like(rate)).min()
            p = x + torch.clamp(t, max=1.0) * step
        d = p - x
        gd = float(g @ d) if g is not None else 0.0
        return p, d, gd, float(d.abs().max())


def solve_budget_capped(plan, costs, cap_rows, cap_rhs, budget, s_norm, prm, x0=None):
    """min_m  || (V_o(m)/s_o)_o ||_p  s.t.  cost.m <= budget, m >= 0, cap_rows[i].m <= cap_rhs[i]; returns (m, info)"""
    dev = plan.device
    L, n_out = plan.L, plan.n_out
    w = np.asarray(costs, dtype=np.float64)
    B = float(budget)
    scale_h = B / w
    scale = torch.from_numpy(np.concatenate([scale_h, [0.0]])).to(dev)          # slack column contributes no samples
    A = torch.from_numpy(np.stack([np.concatenate([np.asarray(r, dtype=np.float64) * scale_h, [0.0]]) for r in cap_rows])).to(dev)
    c = torch.from_numpy(np.asarray(cap_rhs, dtype=np.float64)).to(dev)
    floor = float(prm["scaling_floor"])
    cs = CappedSimplex(A, c, floor)
    s_t = torch.from_numpy(np.asarray(s_norm, dtype=np.float64)).to(dev)
    st = {"norm": 1.0, "p": 32.0, "fevals": 0, "gevals": 0}

    def objective(var):
        r = var / s_norm
        if not np.isfinite(r).all():
            return np.inf, None
        rmax = r.max()
        p = st["p"]
        if np.isinf(p) or n_out == 1:
            coef = np.zeros(n_out)
            coef[int(np.argmax(r))] = 1.0
            return rmax, coef / s_norm
        t = (r / rmax) ** p
        return rmax * t.sum() ** (1.0 / p), (r / rmax) ** (p - 1) * t.sum() ** (1.0 / p - 1.0) / s_norm

    def evaluate(x, want_grad):
        var, grad, status = plan.eval(scale[:L] * x[:L], want_grad=want_grad)
        ok = (status[0].cpu().numpy() == EVAL_OK).all()
        Fv, coef = objective(var[0].cpu().numpy()) if ok else (np.inf, None)
        return Fv, coef, grad

    def feval(x):
        st["fevals"] += 1
        return evaluate(x, False)[0] / st["norm"]

    def geval(x):
        st["gevals"] += 1
        Fv, coef, grad = evaluate(x, True)
        if coef is None:
            raise RuntimeError("capped SPG: gradient requested at an infeasible point")
        g = plan.combine_grad(grad, torch.from_numpy(coef / st["norm"]).to(dev).reshape(1, -1), scale=scale[:L].contiguous())[0]
        return torch.cat([g, torch.zeros(1, dtype=torch.float64, device=dev)])

    def proj(x):
        return cs.project(x)[0]

    def proj_step(x, g, lmbda):
        _, d, gd, dmax = cs.project(x, g, lmbda)
        return d, gd, dmax

    def metric_dot(s_, x_):
        return float((s_ * s_ / torch.clamp(x_, min=floor)).sum())

    if x0 is None:
        x = torch.full((L + 1,), 1.0 / (L + 1), dtype=torch.float64, device=dev)
    else:
        xs = np.asarray(x0, dtype=np.float64) * w / B
        x = torch.from_numpy(np.concatenate([xs, [max(0.0, 1.0 - xs.sum())]])).to(dev)
    x = proj(x)
    F0 = feval(x)
    if not np.isfinite(F0):
        return None, {"reason": "the projected starting point does not sample model 0"}
    st["norm"] = F0
    stages = [float(q) for q in prm["smoothing_p"]] if isinstance(prm["smoothing_p"], (list, tuple)) else [float(prm["smoothing_p"])]
    stages = stages if n_out > 1 else [np.inf]
    tot_it = tot_count = 0
    res = None
    for q in stages:
        st["p"] = q
        res = spg(feval, geval, proj, x, eps=prm["eps"], maxit=max(1, int(prm.get("capped_maxit", 600)) // len(stages)),
                  max_fevals=prm["max_fevals"],
                  verbose=False, lmbda_min=prm["lmbda_min"], lmbda_max=min(float(prm["lmbda_max"]), float(prm.get("capped_lmbda_max", 1.0))),
                  Hlength=prm["linesearch_history_length"],
                  proj_step=proj_step, metric_dot=metric_dot if floor > 0 else None)
        x = res["x"]
        tot_it += res["it"]
        tot_count += res["count"]
    xs = x[:L].cpu().numpy()
    xs[xs < 1.0e-12 * xs.max()] = 0.0
    m = scale_h * xs
    info = {"it": tot_it, "count": tot_count, "gpmax": res["gpmax"], "f": res["f"] * st["norm"], "solver_info": res["solver_info"],
            "fevals": st["fevals"], "gevals": st["gevals"], "pruned": int((xs == 0).sum()), "unspent_budget_share": float(x[L]),
            "projection_launches": cs.launches, "projection_fallbacks": getattr(cs, "fallbacks", 0)}
    return m, info


def solve_capped(plan, costs, cap_rows, cap_rhs, budget=None, eps=None, x0=None, prm=None, unconstrained_cost=None, budget_solver=None):
    """budget mode: one capped solve.  eps mode (min cost s.t. V_o <= eps_o^2): V's homogeneity no longer turns it into a budget
    problem (the caps are absolute), so the smallest budget whose capped optimum meets the tolerances is bracketed and bisected."""
    n_out = plan.n_out

    def one_budget(B, s_norm_, start):
        """one capped solve at budget B: the second-order finish with the caps in its master problem when the caller supplied it
        (budget_solver) and it succeeds, else the first-order loop over the capped set below"""
        if budget_solver is not None:
            got = budget_solver(B, s_norm_, start)
            if got is not None:
                return got
        return solve_budget_capped(plan, costs, cap_rows, cap_rhs, B, s_norm_, prm, x0=start)

    if budget is not None:
        return one_budget(budget, np.ones(n_out), x0)
    s_norm = np.asarray(eps, dtype=np.float64) ** 2

    def attempt(B, start):
        m, info = one_budget(B, s_norm, start)
        return m, info, (np.inf if m is None else info["f"])          # f = max_o V_o/eps_o^2 (p-norm for smooth stages)

    def true_ratio(m):
        var, _, status = plan.eval(m, want_grad=False)
        if not (status[0].cpu().numpy() == EVAL_OK).all():
            return np.inf
        return float((var[0].cpu().numpy() / s_norm).max())

    # Without caps the cost scales like 1/ratio (V is homogeneous of degree -1); with caps that is only a guess, but a good one:
    # B <- B * ratio until the tolerance is met (bracket), then a few bisection steps in log B.  Each step is one capped solve
    # warm-started from the previous allocation.
    lo, m_lo = None, None
    B = float(unconstrained_cost)                        # without caps this budget gives ratio 1: a lower bound on the cost
    m_hi = info_hi = hi = None
    start = x0
    for _ in range(8):
        m_try, info_try, _ = attempt(B, start)
        rho = np.inf if m_try is None else true_ratio(m_try)
        if m_try is not None and rho <= 1.0 + 1.0e-6:
            hi, m_hi, info_hi = B, m_try, info_try
            break
        lo = B
        start = m_try if m_try is not None else start
        B *= min(4.0, max(1.05, rho if np.isfinite(rho) else 4.0))
    if m_hi is None:
        return None, {"reason": "the sample caps make the error tolerance unreachable"}
    if lo is not None:
        for _ in range(6):
            if hi / lo - 1.0 < 2.0e-3:
                break
            mid = float(np.sqrt(lo * hi))
            m_try, info_try, _ = attempt(mid, m_hi)
            if m_try is not None and true_ratio(m_try) <= 1.0 + 1.0e-6:
                hi, m_hi, info_hi = mid, m_try, info_try
            else:
                lo = mid
    return m_hi, info_hi
