"""Prototype: SQP finish on the final support + KKT pricing (tuned: loose pricing tolerance, few entering groups, short SQP runs).
For each shape: default solve -> finish; objective before/after, time, support sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scipy.optimize import minimize
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
from bluest_amd.plan import EVAL_OK
from bluest_amd.sap import support_multipliers


def evaluate(plan, m_h):
    var, grad, status = plan.eval(m_h)
    if not (status[0].cpu().numpy() == EVAL_OK).all():
        return None, None
    r = var[0].cpu().numpy()
    return (r, plan.output_gradients(grad[0])) if np.isfinite(r).all() else (None, None)


def sqp(sub, sc, xk, maxiter):
    n, r0 = len(xk), evaluate(sub, sc * xk)[0]
    F0, n_out, cache = float(r0.max()), len(r0), {}

    def both(z):
        key = z[:-1].tobytes()
        if cache.get("k") != key:
            r, G = evaluate(sub, sc * np.maximum(z[:-1], 0.0))
            cache["k"], cache["v"] = key, ((None, None) if r is None else (r / F0, G * sc[None, :] / F0))
        return cache["v"]
    cons = lambda z: z[-1] - (both(z)[0] if both(z)[0] is not None else np.full(n_out, 1e6))
    def jac(z):
        J = np.zeros((n_out, n + 1)); G = both(z)[1]
        if G is not None: J[:, :-1] = -G
        J[:, -1] = 1.0
        return J
    et = np.zeros(n + 1); et[-1] = 1.0
    ex = np.ones(n + 1); ex[-1] = 0.0
    res = minimize(lambda z: z[-1], np.concatenate([xk / xk.sum(), [1.0]]), jac=lambda z: et, method="SLSQP",
                   constraints=[{"type": "ineq", "fun": cons, "jac": jac}, {"type": "eq", "fun": lambda z: z[:-1].sum() - 1.0, "jac": lambda z: ex}],
                   bounds=[(0.0, 1.0)] * n + [(0.0, 4.0)], options={"maxiter": maxiter, "ftol": 1e-13})
    xn = np.maximum(res.x[:-1], 0.0); xn[xn < 1e-13] = 0.0; xn /= xn.sum()
    rn = evaluate(sub, sc * xn)[0]
    return (xn, float(rn.max()), res.nit) if rn is not None and rn.max() <= F0 else (xk / xk.sum(), F0, res.nit)


def finish(mos, m, B, w, rounds=3, tol=1e-5, maxiter=60):
    L, N = mos.L, mos.N
    sc = B / w
    x = m * w / B
    keep = np.flatnonzero(x > 0)
    info = []
    for rnd in range(rounds):
        sub = mos._restricted_plan(keep)
        xk, f, nit = sqp(sub, sc[keep], np.maximum(x[keep], 1e-12), maxiter)
        x = np.zeros(L); x[keep] = xk
        xi = (1 - 1e-3) * x + 1e-3 / L
        r, G = evaluate(mos.plan, sc * xi)
        act = np.flatnonzero(r >= r.max() * (1 - 1e-3))
        Gs = G[act] * sc[None, :]
        sup = np.flatnonzero(x > 0)
        mu = support_multipliers(Gs[:, sup], xi[sup])
        g = mu @ Gs
        theta = float(g[sup] @ xi[sup]) / xi[sup].sum()
        viol = g - theta; viol[keep] = 0.0
        enter = np.flatnonzero(viol < -tol * abs(theta))
        info.append((len(keep), nit, len(enter)))
        if len(enter) == 0:
            break
        enter = enter[np.argsort(viol[enter])[:N]]
        keep = np.sort(np.concatenate([keep[x[keep] > 0], enter]))
    return x * B / w, info


cfgs = [(16, 5, 2), (16, 5, 4), (18, 5, 3), (20, 4, 2), (20, 4, 8), (22, 4, 4), (22, 5, 2), (14, 6, 3), (24, 4, 2), (20, 5, 8), (20, 5, 1)]
for n, k, o in cfgs:
    prob = synth.problem(n, k, o)
    g = prob["groups"]
    mos = MOSAP(prob["C"], k, [k] * o, [a.copy() for a in g], [[a.copy() for a in g] for _ in range(o)], prob["costs"], [prob["costs"]] * o, verbose=False)
    B, w = prob["budget"], prob["costs"]
    m = mos.solve(budget=B, solver="spg", continuous_relaxation=True)
    V0 = max(mos.variances(m))
    for rep in range(2):
        t0 = time.perf_counter()
        m2, info = finish(mos, m, B, w)
        dt = (time.perf_counter() - t0) * 1e3
    V1 = max(mos.variances(m2))
    print("n=%d k=%d o=%d: V %.9e -> %.9e (%.1e lower) in %.0f ms, rounds (support, sqp its, entering) %s" % (n, k, o, V0, V1, 1 - V1 / V0, dt, info), flush=True)
