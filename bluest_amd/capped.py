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


def cost_shift_capped(plan, costs, s_norm, B, cap_rows, cap_rhs, prm=None, x_start=None, log=None):
    """max_model_samples through the UNCAPPED second-order finish (bluest_amd.colgen.colgen_solve): at the optimum of

        min_m max_o V_o(m)/s_o   s.t.  w.m = B,  a_c.m <= n_c  (a_c = indicator of the groups containing the capped model c)

    stationarity reads  -dF = lambda (w + sum_c theta_c a_c),  theta_c = nu_c / lambda >= 0 -- the capped optimum is the FREE
    optimum under the shifted costs  w' = w + sum_c theta_c a_c  and the budget  B' = B + sum_c theta_c n_c.  So the caps never enter
    the master problem: an outer iteration runs free solves for shifts theta >= 0.

    At the optimal shifts the free problem has a whole FACE of optima and the free solver returns some point of it, so the cap
    violations jump and no shift makes them vanish; the capped optimum is a convex combination on that face.  Hence a
    Dantzig-Wolfe scheme whose columns are whole allocations: (1) probes / a bracketed Broyden iteration on the relative
    violations until the iterates lie on both sides of every active cap; (2) a small LP over the iterates recovers a feasible
    combination (`recover`), whose value is re-evaluated exactly, and the LP's prices of the budget and cap rows are the next
    shifts; (3) when the prices settle on shifts already tried, neighbours a few per cent away are visited (other points of the
    face).  The bound is free: {w.m = B, a_c.m <= n_c} lies inside {w'.m <= B'}, so the certified lower bound of EVERY free solve
    bounds the capped problem, and the iteration stops when the recovered value is within gap_tol of the best bound.
    Works over sharded plans too (nothing but costs and budget changes).  Returns (m, info) or (None, reason)."""
    import time
    from .colgen import colgen_solve
    prm = dict(prm or {})
    w = np.asarray(costs, dtype=np.float64)
    A = np.asarray(cap_rows, dtype=np.float64)
    n = np.asarray(cap_rhs, dtype=np.float64)
    ncap = len(n)
    tol = float(prm.pop("cap_tol", 1.0e-8))
    max_solves = int(prm.pop("cap_max_solves", 24))
    warm_ma = int(prm.pop("cap_warm_ma_iterations", 40))
    warm = bool(prm.pop("cap_warm_start", False))      # measured: a free solve from the previous (sparse) optimum needs more pricing rounds
                                                        # than one from the uniform point (75 vs 20 ms at the headline size)
    state = {"solves": 0, "newton_it": 0, "evals": 0, "rounds": 0}

    def run(theta, x0):
        w2 = w + theta @ A
        B2 = float(B + theta @ n)
        p = dict(prm)
        if x0 is not None and warm:
            p["ma_iterations"] = warm_ma
        t0_ = time.perf_counter()
        x, info = colgen_solve(plan, w2, s_norm, B2, x0=x0 if warm else None, prm=p)
        if x is None:
            return None, info
        if log:
            log("      %.1f ms (ma %.1f, rounds %.1f): rounds %d newton %d full %d master evals %d" % ((time.perf_counter() - t0_) * 1e3, info.get("t_ma_ms", 0), info.get("t_rounds_ms", 0), info["rounds"], info["newton_it"], info["full_evals"], info["master_evals"]))
        state["solves"] += 1
        state["newton_it"] += info["newton_it"]
        state["evals"] += info["full_evals"] + info["master_evals"]
        state["rounds"] += info["rounds"]
        m = (B2 / w2) * x
        g = (A @ m) / n - 1.0
        if log:
            log("cost shift solve %2d: theta %s  violation %s  F %.12e gap %.1e" % (state["solves"], np.array2string(theta, precision=4), np.array2string(g, precision=3), info["F"], info["gap"]))
        return (x, m, g, info), None

    theta = np.zeros(ncap)
    cur, why = run(theta, x_start)
    if cur is None:
        return None, why
    best_lb = cur[3]["lower_bound"]                     # theta = 0: the free problem is a relaxation as well
    active = cur[2] > tol
    lo, hi = np.zeros(ncap), np.full(ncap, np.inf)      # per cap: largest shift seen with the cap violated, smallest seen with it slack
    J = None
    th_prev = g_prev = None

    def bracket(th, g_):
        for c in range(ncap):
            if g_[c] > 0.0:
                lo[c] = max(lo[c], th[c])
                if th[c] >= hi[c]:
                    hi[c] = np.inf              # the other shifts moved: the old observation no longer holds
            elif th[c] > 0.0:
                hi[c] = min(hi[c], th[c])
                if th[c] <= lo[c]:
                    lo[c] = 0.0

    def into_bracket(th_new, th_old, act, probe, g_now=None):
        """usage as a function of one shift is steep and convex (a cheap model drops out altogether): a step that leaves the
        bracket of its cap is replaced by the bracket's geometric middle (or a tenth of the slack end while nothing smaller is known).
        A bracket that has collapsed while the cap is still violated on its lower side has located a JUMP of the usage (two
        vertices of the optimal face): step just across it, so that the recovery sees both vertices at the current shifts"""
        for i, c in enumerate(act):
            if g_now is not None and lo[c] > 0.0 and np.isfinite(hi[c]) and hi[c] / lo[c] - 1.0 < 1.0e-2 and abs(g_now[c]) > 1.0e-3:
                th_new[c] = hi[c] * (1.0 + 2.0e-3) if g_now[c] > 0.0 else lo[c] * (1.0 - 2.0e-3)
                continue
            t = th_new[c]
            if lo[c] < t < hi[c]:
                continue
            if np.isfinite(hi[c]):
                th_new[c] = np.sqrt(lo[c] * hi[c]) if lo[c] > 0.0 else 0.1 * hi[c]
            else:
                th_new[c] = max(2.0 * th_old[c], probe[i])
        return th_new

    bracket(theta, cur[2])
    iterates = []                                       # (m, F, w.m, A m) of every free solve: material for the primal recovery
    recovered = None
    gap_tol = float(prm.get("gap_tol", 1.0e-7))

    def recover():
        """At the optimal shifts the free problem has many optima (a face) and the solver returns one of them, so the violations
        jump and no shift makes them vanish.  The capped optimum is a convex combination on that face: by convexity of F,
        F(sum_j alpha_j m_j) <= sum_j alpha_j F(m_j), so the best combination of the iterates that respects budget and caps (a tiny
        LP in alpha) is a feasible point whose value is re-evaluated exactly and compared with the best lower bound."""
        from scipy.optimize import linprog
        k = len(iterates)
        Fj = np.array([it[1] for it in iterates])
        rows = np.vstack([np.array([it[2] for it in iterates])[None, :] / B, np.stack([it[3] for it in iterates], axis=1) / n[:, None]])
        # An iterate with slack caps overspends the true budget (it spends B' on w'), one with violated caps underspends it: a
        # combination is scaled by t = its largest constraint ratio, which multiplies F by t.  First-order model of
        # t(alpha) * sum_j alpha_j F_j around (1, F_min):  minimise  t + sum_j alpha_j F_j / F_min  -- an LP in (alpha, t).
        cost = np.concatenate([Fj / Fj.min(), [1.0]])
        A_ub = np.hstack([rows, -np.ones((ncap + 1, 1))])
        res = linprog(cost, A_ub=A_ub, b_ub=np.zeros(ncap + 1), A_eq=np.concatenate([np.ones(k), [0.0]])[None, :], b_eq=[1.0],
                      bounds=[(0, None)] * k + [(0, None)], method="highs")
        if res.status != 0:
            return None
        y = -np.asarray(res.ineqlin.marginals)           # prices of the budget row and the cap rows (sum to 1)
        state["lp_theta"] = (y[1:] / n) / (y[0] / B) if y[0] > 1.0e-12 else None
        alpha = np.maximum(res.x[:k], 0.0)
        m_c = sum(a_ * it[0] for a_, it in zip(alpha, iterates) if a_ > 0)
        t_ = max(float(w @ m_c) / B, float(((A @ m_c) / n).max()))
        m_c = m_c / t_                                   # exactly feasible; spends the whole budget or sits at a cap
        var, _, status = plan.eval(m_c, want_grad=False)
        if not (status[0].cpu().numpy() == EVAL_OK).all():
            return None
        state["evals"] += 1
        return m_c, float((var[0].cpu().numpy() / np.asarray(s_norm)).max())

    while True:
        x, m, g, info = cur
        best_lb = max(best_lb, info["lower_bound"])
        iterates.append((m, info["F"], float(w @ m), A @ m))
        viol = np.where(active, np.abs(g), np.maximum(g, 0.0))
        newly = (~active) & (g > tol)
        if viol.max() <= tol:
            break
        if len(iterates) >= 3 and hasattr(plan, "eval"):
            got = recover()
            if got is not None and (recovered is None or got[1] < recovered[1]):
                recovered = got
            if log and got is not None:
                log("   recovery over %d iterates: F %.12e  gap %.2e" % (len(iterates), got[1], 1.0 - best_lb / got[1]))
            if recovered is not None and 1.0 - best_lb / recovered[1] <= max(gap_tol, 1.0e-7):
                break
        if state["solves"] >= max_solves:
            if viol.max() <= 1.0e-5 or (recovered is not None and 1.0 - best_lb / recovered[1] <= 1.0e-4):
                break
            why = "cost shift did not converge in %d solves (violation %.1e)" % (state["solves"], viol.max())
            if recovered is not None and 1.0 - best_lb / recovered[1] <= 1.0e-3:
                # not good enough to stop looking, good enough to beat an uncertified first-order answer: the caller may take it
                m_c, F_c = recovered
                return None, {"reason": why, "candidate": (m_c, {
                    "F": F_c, "gap": 1.0 - best_lb / F_c, "lower_bound": best_lb, "newton_it": state["newton_it"], "rounds": state["rounds"],
                    "full_evals": state["evals"], "master_evals": 0, "kkt": float(viol.max()), "support": int((m_c > 0).sum()),
                    "cap_usage": ((A @ m_c) / n).tolist(), "mu": info.get("mu"), "solves": state["solves"], "theta": theta.tolist()})}
            return None, why
        active = active | newly
        act = np.flatnonzero(active)
        # Dantzig-Wolfe step: once the iterates lie on both sides of every active cap, the prices of the recovery LP (its budget
        # and cap rows) ARE the next shifts -- column generation over free optima, the columns being whole allocations
        G = np.stack([it[3] for it in iterates], axis=1) / n[:, None] - 1.0
        both = all((G[c] > 0).any() and (G[c] < 0).any() for c in act)
        if both and state.get("lp_theta") is not None and len(iterates) >= 3:
            th_new = np.maximum(np.asarray(state["lp_theta"], dtype=np.float64), 0.0)
            if np.isfinite(th_new).all() and any(np.allclose(th_new, it_th, rtol=1e-3, atol=0) for it_th in state.setdefault("thetas", [])):
                # the prices have settled but the free solver keeps returning the same few points of the optimal face: look at its
                # neighbours -- one shift a few per cent up or down, cap by cap
                k_ = state["perturb"] = state.get("perturb", -1) + 1
                c_ = act[k_ % len(act)]
                th_new[c_] *= 1.0 + (0.03 if (k_ // len(act)) % 2 == 0 else -0.03) * (1 + k_ // (2 * len(act)))
            if np.isfinite(th_new).all() and not any(np.allclose(th_new, it_th, rtol=1e-9, atol=0) for it_th in state.setdefault("thetas", [])):
                state["thetas"].append(th_new.copy())
                nxt, why = run(th_new, x)
                if nxt is None:
                    return None, why
                bracket(th_new, nxt[2])
                th_prev, g_prev, J = None, None, None
                theta, cur = th_new, nxt
                continue
        # scale of a shift: a thousandth of the usage-weighted mean cost of the cap's groups (the capped model is usually a cheap one)
        usage_w = np.array([float((A[c] * m) @ w / max(A[c] @ m, 1e-300)) if (A[c] @ m) > 0 else float((A[c] @ w) / max(A[c].sum(), 1.0)) for c in act])
        probe = 1.0e-3 * usage_w
        never_slack = [c for c in act if not np.isfinite(hi[c]) and g[c] > tol]
        if never_slack and len(iterates) >= 4:
            # a cap that has been violated at every shift tried so far: quadruple its shift until it is not
            th_new = theta.copy()
            for c in never_slack:
                th_new[c] = max(4.0 * theta[c], probe[list(act).index(c)])
            nxt, why = run(th_new, x)
            if nxt is None:
                return None, why
            bracket(th_new, nxt[2])
            th_prev, g_prev, J = None, None, None
            theta, cur = th_new, nxt
            continue
        if newly.any() or J is None or J.shape[0] != len(act):
            # (re)start: one probe along all active caps, diagonal secant
            step = np.where(theta[act] > 0, 0.1 * theta[act], probe) * np.where(g[act] >= 0, 1.0, -1.0)
            th_new = theta.copy()
            th_new[act] = np.maximum(theta[act] + step, 0.0)
            th_new = into_bracket(th_new, theta, act, probe)
            nxt, why = run(th_new, x)
            if nxt is None:
                return None, why
            bracket(th_new, nxt[2])
            dth = (th_new - theta)[act]
            dg = nxt[2][act] - g[act]
            diag = np.where((np.abs(dg) > 1e-12) & (np.abs(dth) > 0), dg / np.where(dth != 0, dth, 1.0), -1.0 / np.maximum(probe, 1e-300))
            J = np.diag(np.minimum(diag, -1e-6 / np.maximum(usage_w, 1e-300)))       # more cost, less usage
            th_prev, g_prev = theta.copy(), g.copy()
            theta, cur = th_new, nxt
            continue
        if th_prev is not None:                                   # Broyden update on the active set
            dth, dg = (theta - th_prev)[act], (g - g_prev)[act]
            den = float(dth @ dth)
            if den > 0:
                J = J + np.outer(dg - J @ dth, dth) / den
        try:
            step = -np.linalg.solve(J, g[act])
        except np.linalg.LinAlgError:
            step = -g[act] / np.minimum(np.diag(J), -1e-300)
        if not np.isfinite(step).all():
            step = -g[act] / np.minimum(np.diag(J), -1e-300)
        th_new = theta.copy()
        th_new[act] = np.maximum(theta[act] + step, 0.0)
        th_new = into_bracket(th_new, theta, act, probe, g)
        drop = (th_new[act] <= 0.0) & (g[act] < 0.0)              # a cap that is slack at zero shift leaves the active set
        th_prev, g_prev = theta.copy(), g.copy()
        nxt, why = run(th_new, x)
        if nxt is None:
            return None, why
        bracket(th_new, nxt[2])
        if drop.any():
            active[act[drop]] = False
            J = None if not active.any() else J[np.ix_(~drop, ~drop)]
            th_prev = None
        theta, cur = th_new, nxt
    x, m, g, info = cur
    # feasible point: one common factor (V is homogeneous of degree -1, so F grows by exactly that factor)
    t = max(1.0, float(w @ m) / B, float(((A @ m) / n).max()))
    m = m / t
    F = info["F"] * t
    if recovered is not None and recovered[1] <= F:
        m, F = recovered
    out = {"F": F, "gap": 1.0 - best_lb / F, "lower_bound": best_lb, "newton_it": state["newton_it"], "rounds": state["rounds"],
           "full_evals": state["evals"], "master_evals": 0, "kkt": float(np.where(theta > 0, np.abs(g), np.maximum(g, 0.0)).max()),
           "support": int((m > 0).sum()), "cap_usage": ((A @ m) / n).tolist(), "mu": info.get("mu"), "solves": state["solves"],
           "theta": theta.tolist()}
    return m, out

