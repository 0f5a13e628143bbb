import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import oracle
from oracle.master_newton import SupportProblem, master_newton, dual_bound
from bluest_amd import synth

def colgen(saps, w, B, s, rounds=80, enter_per=None, eps_list=(1e-3, 1e-6, 1e-9), tol=1e-8, verbose=True, vm=False, init_mult=2, adaptive=True, ma_its=0, ma_p=32.0):
    n_out = len(saps); sp0 = saps[0]; N, L = sp0.N, sp0.L
    c = B / w
    flatg = [g for gk in sp0.groups for g in gk]
    def block(o, i):
        k = int(np.searchsorted(sp0.cumsizes, i, side='right'))
        j = i - sp0.cumsizes[k-1]
        return saps[o].invcovs[k-1].reshape(-1, k, k)[j]
    def full_eval(x):
        m = c * x
        out = [q.variance_GH(m) for q in saps]
        return np.array([v for v,_,_ in out]), np.array([g for _,g,_ in out])
    if enter_per is None: enter_per = N
    u = np.full(L, 1.0 / L)
    PHIu = np.array([q.get_phi(c * u) for q in saps])
    V, Gm = full_eval(u)
    r = V / s
    mu = np.where(r >= r.max() * (1 - 1e-3), 1.0, 0.0); mu /= mu.sum()
    ci = c * ((mu / s) @ (-Gm))
    keep = np.union1d(np.argsort(-ci)[:init_mult * N], [0])
    xS = np.full(len(keep), 1.0 / len(keep))
    if ma_its:
        xm = u.copy()
        for t in range(ma_its):
            V, Gm = full_eval(xm)
            r = V / s
            wgt = (r / r.max()) ** (ma_p - 1); wgt /= wgt.sum()      # gradient weights of the p-norm
            gx = c * ((wgt / s) @ (-Gm))                              # -d/dx of sum_o wgt_o r_o   (> 0)
            xm = xm * gx / float(wgt @ r)                             # sum_i x_i gx_i = sum_o wgt_o r_o (homogeneity): stays on the simplex
            xm /= xm.sum()
        print("MA: F after %d its = %.9e" % (ma_its, (full_eval(xm)[0] / s).max()))
        keep = np.union1d(np.argsort(-xm)[:init_mult * N], [0])
        xS = xm[keep] / xm[keep].sum()
        mu = wgt
    nevals = 1; t_master = 0; tot_newton = 0; tot_mevals = 0; nr = 0; mtol = 1e-2
    for eps_bg in eps_list:
      for rnd in range(rounds):
        nr += 1
        prob = SupportProblem(N, [flatg[i] for i in keep], [[block(o, i) for i in keep] for o in range(n_out)], c[keep], s, eps_bg * PHIu, eps_bg)
        t0 = time.time()
        res = master_newton(prob, xS, mu0=mu, verbose=vm, tol=mtol if adaptive else 1e-9)
        t_master += time.time() - t0
        tot_newton += res["it"]; tot_mevals += res["evals"]
        xS, mu = res["x"], res["mu"]
        x = np.zeros(L); x[keep] = xS
        F = res["F"]
        xi = (1 - eps_bg) * x + eps_bg * u
        V, Gm = full_eval(xi); nevals += 1
        assert abs((V / s).max() / F - 1) < 1e-6, ((V/s).max(), F)
        LB, ci = dual_bound(-Gm, V, mu, s, c)
        gap = 1 - LB / F
        pos = xS > 0
        level = float(ci[keep][pos] @ xS[pos]) / xS[pos].sum()
        viol = ci / level - 1
        viol[keep[pos]] = -1
        enter = np.argsort(-viol)[:enter_per]
        enter = enter[viol[enter] > tol]
        Fsparse = max(q.variance(c * x) for q in saps) if verbose else 0
        if verbose: print("eps %.0e round %2d F_eps %.12e F(x) %.12e gap %.3e |S| %3d nnz %3d newton it %2d evals %3d enter %3d maxviol %.2e kkt %.1e mu %s" % (eps_bg, rnd, F, Fsparse, gap, len(keep), int(pos.sum()), res["it"], res["evals"], len(enter), float(viol.max()), res["kkt"], np.round(mu,4)))
        if len(enter) == 0:
            if mtol > 1e-9 and adaptive:
                mtol = 1e-9; continue
            break
        if adaptive: mtol = max(1e-9, min(1e-2, 1e-2 * float(viol.max())))
        keep_new = np.concatenate([keep[pos], enter])
        x_new = np.concatenate([xS[pos], np.zeros(len(enter))])      # enter at zero: the master frees them (rc < 0)
        order = np.argsort(keep_new)
        keep, xS = keep_new[order], x_new[order] / x_new.sum()
    return x, mu, F, gap, dict(rounds=nr, newton=tot_newton, master_evals=tot_mevals, full_evals=nevals, t_master=t_master)

if __name__ == "__main__":
    n, kmax, n_out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    prob = synth.problem(n, kmax, n_out)
    saps = [oracle.SparseOracleSAP(C, kmax, prob["groups"]) for C in prob["C"]]
    t0 = time.time()
    kw = dict(a.split("=") for a in sys.argv[4:])
    kw = {k: eval(v) for k, v in kw.items()}
    x, mu, F, gap, st = colgen(saps, prob["costs"], prob["budget"], np.ones(n_out), verbose=False, **kw)
    c = prob["budget"] / prob["costs"]
    print("total", time.time() - t0, st, "F", max(q.variance(c * x) for q in saps), "gap", gap, "nnz", (x > 0).sum())
