"""
oracle/master_newton.py -- CPU restatement (numpy) of the build-defined SECOND-ORDER FINISH of solver="spg":
column generation over the groups + an active-set Newton (SQP) method on the current support.

TEST INFRASTRUCTURE ONLY (like the rest of oracle/): the product runs the HIP kernels of bluest_amd/csrc/newton.hip; this file
states the same algorithm in numpy so that the kernels can be checked step by step, and it is where the algorithm was
developed.  The reference has no counterpart: its NLP back-ends (bluest/sap.py:387-456 scipy trust-constr / ipopt with the
Hessian of bluest/misc.py:497-503, bluest/cmisc.cpp:74-97) are third-party solvers.

Problem (bluest/sap.py:387-418, bluest/mosap.py:578-605 in the scaled variable x_i = w_i m_i / B on the unit simplex):

    min_x  F(x) = max_o r_o(x),   r_o = V_o(m(x)) / s_o,   V_o = e_0^T Phi_o(m)^-1 e_0 on the sampled models,
    s.t.   sum x = 1,  x >= 0   (+ optional caps  A_c x <= b_c:  max_model_samples, bluest/sap.py:222-240)

Derivatives on a support S (a_{o,i} = C_{i,o}^-1 v_o[g_i] scattered to the models of group i, c_i = B / w_i, T_o = Phi_o^-1):
    dr_o/dx_i        = -c_i a_{o,i}.v_o / s_o
    d2r_o/dx_i dx_j  = 2 c_i c_j a_{o,i}^T T_o a_{o,j} / s_o            (bluest/misc.py:497-503 restricted to S)

Master problem on S: SQP in epigraph form with the outputs within act_tol of the maximum as equality constraints
(r_o + g_o.d = tau), multipliers mu >= 0 (an output whose multiplier comes out negative leaves the active set), the
Lagrangian Hessian plus a proximal term rho * diag(1 / max(x, floor)), entries at zero with a non-negative reduced cost held
fixed, step to the boundary + Armijo backtracking on the true max.

Pricing + certificate: for ANY vectors y_o and ANY mu in the simplex every feasible allocation has
    F >= LB = A^2 / (4 max_i c_i),   A = 2 sum_o (mu_o / s_o) y_{o,0},   c_i = (B / w_i) sum_o (mu_o / s_o) y_{o,g_i}^T C_{i,o}^-1 y_{o,g_i}
(weak duality, see oracle.optimality_certificate; in the scaled variable the budget is 1).  With y_o = v_o at a slightly
interior point and the master's multipliers, c_i is the reduced cost the column generation prices with, so every pricing round
yields a certified gap (F - LB) / F for free; groups with c_i above the support's level enter.
"""
import numpy as np


class SupportProblem(object):
    """the data the master needs: for every output the k x k pseudo-inverse blocks of the groups in the support, and the
    BACKGROUND information matrices bg[o] = eps * Phi_o(uniform allocation over ALL groups): the master minimises
    F_eps(x) = F((1 - eps) x + eps u).  With the background every model is sampled, so F_eps is smooth on the whole simplex (V
    itself has kinks where a model drops out: bluest/misc.py:467-470 restricts Phi to the sampled models) and the reduced costs
    the column generation prices with are the exact gradient of the function the master minimises."""

    def __init__(self, N, groups_S, blocks_S, c_S, s, bg, eps_bg):
        """groups_S: list of index arrays (models of each support group); blocks_S[o][j]: (k_j, k_j) block of output o or None
        when output o does not use group j; c_S: B / w_j; s: per-output scale; bg: (n_out, N, N)"""
        self.N, self.groups, self.blocks, self.c, self.s = N, groups_S, blocks_S, np.asarray(c_S, dtype=np.float64), np.asarray(s, dtype=np.float64)
        self.n_out, self.S = len(blocks_S), len(groups_S)
        self.bg, self.eps_bg = np.asarray(bg, dtype=np.float64), float(eps_bg)

    def evaluate(self, x, want_derivatives=False, mu=None):
        """r (n_out,), and with want_derivatives: G (S, n_out) gradients, Hs list of (S, S) Hessians of the outputs with
        mu_o > 0 (None otherwise), v (n_out, N)"""
        N, S, n_out = self.N, self.S, self.n_out
        m = (1.0 - self.eps_bg) * self.c * x
        r = np.full(n_out, np.inf)
        G = np.zeros((S, n_out))
        Hs = [None] * n_out
        vs = np.zeros((n_out, N))
        for o in range(n_out):
            PHI = self.bg[o].copy()
            for j in range(S):
                if self.blocks[o][j] is None or m[j] <= 0.0:
                    continue
                g = self.groups[j]
                PHI[np.ix_(g, g)] += m[j] * self.blocks[o][j]
            idx = np.flatnonzero(np.diag(PHI) > 0.0)               # models nobody samples stay out (bluest/misc.py:467-470)
            if len(idx) == 0 or idx[0] != 0:
                continue
            try:
                T = np.zeros((N, N))
                T[np.ix_(idx, idx)] = np.linalg.inv(PHI[np.ix_(idx, idx)])
            except np.linalg.LinAlgError:
                continue
            if not np.isfinite(T).all() or T[0, 0] <= 0.0:
                continue
            v = T[:, 0]
            r[o] = T[0, 0] / self.s[o]
            vs[o] = v
            if not want_derivatives:
                continue
            A = np.zeros((N, S))                                    # columns a_{o,j}
            for j in range(S):
                if self.blocks[o][j] is None:
                    continue
                g = self.groups[j]
                A[g, j] = self.blocks[o][j] @ v[g]
            cc = (1.0 - self.eps_bg) * self.c
            G[:, o] = -cc * (A.T @ v) / self.s[o]
            if mu is not None and mu[o] > 0.0:
                W = A * cc[None, :]
                Hs[o] = (2.0 / self.s[o]) * (W.T @ T @ W)
        return (r, G, Hs, vs) if want_derivatives else r


def master_newton(prob, x0, mu0=None, tol=1.0e-9, maxit=60, act_tol=1.0e-3, floor=1.0e-6, verbose=False, fb=None, caps=None, nu0=None):
    """active-set Newton (SQP) on the support with Levenberg-Marquardt damping.  Works on rho_o = -1 / r_o (convex as well:
    1 / V_o is the Schur complement of Phi_o, concave and homogeneous of degree +1 in x; same minimisers; Newton does not crawl
    on it far from the optimum the way it does on the degree -1 function r_o, where a step is x -> 1.5 x).
    Damping: M = H + damp * |lam| * diag(1 / max(x, floor)) -- in units of the multiplier lam, damp = 1 is a multiplicative
    (mirror-descent like) step, damp -> 0 the Newton step; damp follows the ratio actual / predicted decrease, and a rejected
    trial point raises it and re-solves (no line search: an entering column sits at x_j = 0 where the quadratic model is only
    valid for steps of the size of the background, so step LENGTH, not step fraction, is what must adapt).
    caps = (Acap (n_caps, S), bcap (n_caps,)): linear rows Acap x <= bcap (max_model_samples, bluest/sap.py:222-240, in the scaled
    variable); x0 must satisfy them.  Caps at their bound (or carrying a multiplier) are equality rows of the SQP step, a cap
    whose multiplier comes out negative leaves; a trial point that would violate a cap is pulled back along the segment from x.
    Returns dict(x, mu, nu, lam, F, it, evals, kkt)"""
    S, n_out = prob.S, prob.n_out
    Acap = np.zeros((0, S)) if caps is None else np.asarray(caps[0], dtype=np.float64).reshape(-1, S)
    bcap = np.zeros(0) if caps is None else np.asarray(caps[1], dtype=np.float64)
    ncap = len(bcap)
    nu = np.zeros(ncap) if nu0 is None else np.maximum(np.asarray(nu0, dtype=np.float64), 0.0)
    MCAP, PACT = 4, 8
    if fb is None:
        fb = 0.0 if prob.eps_bg > 0.0 else 0.9     # without the background V has kinks where a model drops out: stay inside the face
    x = np.maximum(np.asarray(x0, dtype=np.float64), 0.0)
    x = x / x.sum()
    mu = np.full(n_out, 1.0 / n_out) if mu0 is None else np.asarray(mu0, dtype=np.float64).copy()
    damp = 1.0e-2
    info = {"it": 0, "evals": 0, "solves": 0}
    r = prob.evaluate(x)
    info["evals"] += 1
    if not np.isfinite(r.max()):
        raise ValueError("master: the starting point is not evaluable")
    kkt, lam = np.inf, 0.0
    best, fails, in_noise, noise_next, noise_stop = None, 0, False, False, 0
    for it in range(maxit):
        F = r.max()
        # candidate outputs of the step: within act_tol of the maximum or carrying a multiplier, at most PACT (largest first).  They
        # enter the step as equality rows; one whose multiplier comes out negative leaves, and comes back (locked) if the step it
        # was dropped from would lift it above the others to first order -- an active-set solve of the step's QP over the candidates
        cand = np.flatnonzero((r >= F * (1.0 - act_tol)) | (mu > 1.0e-12))
        act0 = np.sort(cand[np.argsort(-r[cand], kind="stable")[:PACT]])
        mu_h = np.zeros(n_out)
        mu_h[act0] = np.maximum(mu[act0], 0.0)
        mu_h = mu_h / mu_h.sum() if mu_h.sum() > 0 else np.where(np.isin(np.arange(n_out), act0), 1.0 / len(act0), 0.0)
        mu_h[act0] = np.maximum(mu_h[act0], 1.0e-3 / len(act0))     # every active output contributes curvature
        mu_h /= mu_h.sum()
        r, G, Hs, _ = prob.evaluate(x, True, mu_h)
        info["evals"] += 1
        # reciprocal form: q_o = -1/r_o, grad = g / r^2, Hess = H / r^2 - 2 g g^T / r^3
        Gq = G / (r ** 2)[None, :]
        q = -1.0 / r
        H = np.zeros((S, S))
        for o in act0:
            H += mu_h[o] * (Hs[o] / r[o] ** 2 - 2.0 * np.outer(G[:, o], G[:, o]) / r[o] ** 3)
        slack = bcap - Acap @ x
        candc = np.flatnonzero((slack <= 1.0e-10 * np.maximum(np.abs(bcap), 1.0)) | (nu > 0.0))
        actc0 = np.sort(candc[np.argsort(slack[candc], kind="stable")[:MCAP]])        # at most MCAP caps in the step at once
        gl = Gq @ mu_h + Acap.T @ nu
        lam_est = -float(gl @ x)
        rc = gl + lam_est
        free = (x > 0.0) | (rc < 0.0)
        fi = np.flatnonzero(free)
        D = 1.0 / np.maximum(x, floor)
        accepted, noise_next = False, False
        for attempt in range(40):
            act = act0.copy()
            actc = actc0.copy()
            locked = set()                                         # caps that came back: dropping them made the step violate them
            olocked = set()                                        # outputs that came back
            while True:                                            # drop outputs / caps whose multiplier comes out negative
                M = H[np.ix_(fi, fi)] + damp * abs(lam_est) * np.diag(D[fi])
                try:
                    Lc = np.linalg.cholesky(M)
                except np.linalg.LinAlgError:
                    damp *= 10.0
                    continue
                info["solves"] += 1
                E = np.column_stack([Gq[fi][:, act], Acap[actc][:, fi].T, np.ones(len(fi))])
                Y = np.linalg.solve(Lc, E)
                K = Y.T @ Y
                p, pc = len(act), len(actc)
                ne = p + pc + 1
                KK = np.zeros((ne + 1, ne + 1))
                KK[:ne, :ne] = K
                KK[:p, ne] = 1.0
                KK[ne, :p] = 1.0
                rhs = np.concatenate([q[act], -slack[actc], [0.0, 1.0]])       # d = -M^-1 E z: (K z)_c = -(b_c - a_c.x)
                try:
                    z = np.linalg.solve(KK, rhs)
                except np.linalg.LinAlgError:
                    z = np.linalg.lstsq(KK, rhs, rcond=None)[0]
                mu_new, nu_new, lam, tau = z[:p], z[p:p + pc], z[p + pc], z[ne]
                odrop = [j for j in range(p) if mu_new[j] < -1.0e-12 and int(act[j]) not in olocked]
                if p > 1 and odrop:
                    act = np.delete(act, min(odrop, key=lambda j: mu_new[j]))
                    continue
                droppable = [j for j in range(pc) if nu_new[j] < -1.0e-12 and int(actc[j]) not in locked]
                if droppable:
                    actc = np.delete(actc, min(droppable, key=lambda j: nu_new[j]))
                    continue
                d = np.zeros(S)
                d[fi] = -np.linalg.solve(Lc.T, Y @ z[:ne])
                # a cap at its bound that was dropped must not be violated by the step it was dropped from
                back = [int(c_) for c_ in actc0 if c_ not in actc and slack[c_] <= 1.0e-10 * max(abs(bcap[c_]), 1.0)
                        and Acap[c_] @ d > 1.0e-12 * max(abs(bcap[c_]), 1.0)]
                if back:
                    locked.update(back)
                    actc = np.sort(np.concatenate([actc, np.asarray(back, dtype=actc.dtype)]))
                    continue
                # a candidate output that was dropped must not rise above the level tau of the others to first order
                oback = [int(o_) for o_ in act0 if o_ not in act and q[o_] + Gq[:, o_] @ d > tau + 1.0e-10 * abs(tau)]
                if oback:
                    olocked.update(oback)
                    act = np.sort(np.concatenate([act, np.asarray(oback, dtype=act.dtype)]))
                    continue
                break
            mu_full = np.zeros(n_out)
            mu_full[act] = np.maximum(mu_new, 0.0)
            nu_full = np.zeros(ncap)
            nu_full[actc] = np.maximum(nu_new, 0.0) / mu_full.sum()
            mu_full /= mu_full.sum()
            if attempt == 0:
                # KKT residual at x with the new multipliers, relative to the level lam
                glx = Gq @ mu_full + Acap.T @ nu_full
                lam_x = -float(glx @ x)
                rcx = glx + lam_x
                pos = x > 1.0e-10                                  # entries below 1e-10 count as at the bound
                kkt = max(float(np.abs(rcx[pos]).max()), float(np.maximum(-rcx[~pos], 0.0).max()) if (~pos).any() else 0.0) / max(abs(lam_x), 1e-300)
                spread = float((F - r[act]).max() / F) if len(act) > 1 else 0.0
                if verbose:
                    print("   newton it %2d F %.12e kkt %.2e spread %.2e |act| %d free %d nnz %d damp %.1e" % (it, F, kkt, spread, len(act), len(fi), pos.sum(), damp))
                if kkt <= tol and spread <= tol:
                    break
                # the best point seen (smallest KKT measure) with the multipliers estimated at it.  Steps taken on trust (see the
                # acceptance test) are judged here: three in a row that do not improve on the best point end the iteration there
                m_now = max(kkt, spread)
                if best is None or m_now < best["m"]:
                    best, fails = {"m": m_now, "x": x, "r": r, "kkt": kkt, "mu": mu_full, "nu": nu_full}, 0
                elif in_noise:
                    fails += 1
                    if fails >= 3:
                        noise_stop = 1
                        break
            pred = (tau - q.max()) * F * F                         # first-order predicted change of F for the step (< 0)
            if pred < -0.5 * F:                                    # the model promises more than half of a positive objective
                damp *= 10.0
                continue
            # projected step: entries that would turn negative become zero; fb > 0 (no background): every entry keeps at least
            # 1 - fb of its value, so that no model drops out of the information matrix inside the master (V has a kink there)
            def pull_back(xv):
                """make a trial point respect every cap (x does).  Clipping negative entries and renormalising moves mass between
                capped and uncapped groups, so a cap the step kept at its bound can end slightly violated: first REPAIR -- scale
                the entries of the most violated cap's groups down to its bound and hand the freed mass to the other entries in
                proportion (a few rounds: caps overlap) --, then, if something is still violated, the furthest feasible point of
                the segment x -> xv"""
                if ncap == 0:
                    return xv
                xv = xv.copy()
                for _ in range(8):
                    ratio = (Acap @ xv) / np.where(bcap > 0, bcap, 1.0)
                    c_ = int(np.argmax(ratio))
                    if ratio[c_] <= 1.0 + 1.0e-13:
                        break
                    msk = Acap[c_] > 0.0
                    rest = float(xv[~msk].sum())
                    if rest <= 0.0:
                        break
                    freed = (1.0 - 1.0 / ratio[c_]) * float(xv[msk].sum())
                    xv[msk] /= ratio[c_]
                    xv[~msk] *= 1.0 + freed / rest
                ax, axv = Acap @ x, Acap @ xv
                bad = axv > bcap * (1.0 + 1.0e-12) + 1e-300
                if not bad.any():
                    return xv
                theta = float(np.min((bcap - ax)[bad] / (axv - ax)[bad]))
                return x + max(theta, 0.0) * (xv - x)
            xt = np.maximum(x + d, (1.0 - fb) * x if fb > 0.0 else 0.0)
            xt = pull_back(xt / xt.sum())
            rt = prob.evaluate(xt)
            info["evals"] += 1
            actual = rt.max() - F
            # below a promised decrease of 1e-11 F the objective no longer tells a good step from a bad one: such a step is taken when
            # the objective does not RISE by more than that, and the next iterations judge it by the KKT residual (best / fails)
            noise = pred > -1.0e-11 * F and max(kkt, spread) <= 1.0e-4 and damp <= 1.0e-2      # (a tiny step of a heavily damped system is not noise)
            ok = np.isfinite(rt.max()) and (actual <= 1.0e-4 * min(pred, 0.0) + 1e-15 * F or (noise and actual <= 1.0e-11 * F))
            if not ok and len(act) > 1 and np.isfinite(rt.max()):
                # near a tie of several outputs second-order errors split the tie and the exact max rejects a good SQP step (the
                # Maratos effect).  Second-order correction: the minimum-norm (in M) step c that re-equalises the active outputs
                # at the trial point to first order,  g_o.c - tau2 = -q_o(xt),  1.c = 0  -- the same K with another right-hand side
                qt = -1.0 / rt[act]
                rhs2 = np.concatenate([qt, np.zeros(pc), [0.0, 0.0]])            # the correction does not move along the active caps
                try:
                    z2 = np.linalg.solve(KK, rhs2)
                except np.linalg.LinAlgError:
                    z2 = None
                if z2 is not None:
                    c = np.zeros(S)
                    c[fi] = -np.linalg.solve(Lc.T, Y @ z2[:ne])
                    xt2 = np.maximum(x + d + c, (1.0 - fb) * x if fb > 0.0 else 0.0)
                    xt2 = pull_back(xt2 / xt2.sum())
                    rt2 = prob.evaluate(xt2)
                    info["evals"] += 1
                    actual2 = rt2.max() - F
                    if np.isfinite(rt2.max()) and (actual2 <= 1.0e-4 * min(pred, 0.0) + 1e-15 * F or (noise and actual2 <= 1.0e-11 * F)):
                        ok, xt, rt, actual = True, xt2, rt2, actual2
            if ok:
                accepted = True
                ratio = actual / pred if pred < 0 else 1.0
                if verbose:
                    print("        accepted at attempt %d damp %.1e: F -> %.12e (pred %.3e actual %.3e)" % (attempt, damp, rt.max(), pred, actual))
                if noise:
                    ratio = 1.0                                    # a step taken on trust counts as a good one
                if ratio > 0.5:
                    damp = max(damp * 0.3, 1.0e-14)      # (x0.1 made every other step a rejected one: profiles/r04_damp_ab.txt)
                elif ratio < 0.1:
                    damp *= 10.0
                noise_next = noise
                break
            if verbose:
                print("        rejected attempt %d damp %.1e pred %.3e actual %.3e rt-F %s" % (attempt, damp, pred, actual, (rt - F)[act]))
            damp *= 10.0
            if damp > 1.0e12:
                break
        if noise_stop:                                             # back to the best point, with its multipliers
            x, r, kkt, mu, nu = best["x"], best["r"], best["kkt"], best["mu"], best["nu"]
            break
        if kkt <= tol and spread <= tol:
            mu, nu = mu_full, nu_full
            break
        if not accepted:
            break
        x, r, mu, nu = xt, rt, mu_full, nu_full
        in_noise = noise_next
        info["it"] = it + 1
    info.update({"x": x, "mu": mu, "nu": nu, "lam": lam, "F": float(r.max()), "r": r, "kkt": kkt})
    return info


def dual_bound(q, y0, mu, s, c):
    """LB = A^2 / (4 max_i c_i) from the quadratic forms q[o][i] = y_{o,g_i}^T C_{i,o}^-1 y_{o,g_i} (any y), y0[o] = y_{o,0},
    mu in the simplex, c_i = B / w_i (budget 1 in the scaled variable).  Returns (LB, c_i vector)"""
    a = np.asarray(mu, dtype=np.float64) / np.asarray(s, dtype=np.float64)
    A = 2.0 * float(a @ np.asarray(y0, dtype=np.float64))
    ci = np.asarray(c, dtype=np.float64) * (a @ np.asarray(q, dtype=np.float64))
    cmax = float(ci.max())
    return (A * A / (4.0 * cmax) if cmax > 0 else 0.0), ci
