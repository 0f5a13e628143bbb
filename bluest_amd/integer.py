"""
Integer projection of a continuous allocation: same search as the reference --
bluest/misc.py:141-167 (bounds), :228-311 (multi-output brute force), :313-382 (single output), with the fallback
ladders of bluest/sap.py:145-187 and bluest/mosap.py:212-289 -- but the heavy step, forming
phis = basephi + psi[:, idx] @ ms for up to 2^LL candidates and taking pinv(phis)[:,0,0] (misc.py:293-294, 368-369),
runs on the GPU (`bluest_intproj_eval`: one wavefront per candidate and output, register-resident Gauss-Jordan solve).

The candidate enumeration and the linear filters (model-0 constraint, budget, ordering) are host numpy on <= 2^20
small integers per chunk, written to keep the reference's selection rule (including its ordering of candidates).
`MOSAP.cleanup_solution` (mosap.py:125-210) supplies the second starting point of the multi-output ladder.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check
from .plan import _stream

CHUNK_BITS = 18   # candidates generated / evaluated per chunk: 2^18


def get_feasible_integer_bounds(sol, N, e=None):
    """bluest/misc.py:141-167: the ~1.2N largest entries (plus the N largest among the groups containing model 0)
    get floor/ceil bounds; everything else is rounded"""
    idx = np.argsort(sol)[-int(1.2 * N):]
    idx = np.array([item for item in idx if sol[item] > 1.0e-8], dtype=np.int64)
    if e is not None:
        if sum(e > 0.99) == 0:
            val = 1 / sum(e) / 2
            while sum(e > val) == 0:
                val /= 2
        else:
            val = 0.99
        idx2 = np.argwhere(e > val).flatten()
        temp = np.argsort(sol[e > val])[::-1]
        idx2 = idx2[temp[:N]]
        idx = np.unique(np.concatenate([idx, idx2]))
    L = len(sol)
    lb = np.zeros((L,), dtype=int)
    ub = np.zeros((L,), dtype=int)
    lb[idx] = np.floor(sol).astype(int)[idx]
    ub[idx] = np.ceil(sol).astype(int)[idx]
    temp = np.argsort(lb[idx])[::-1]
    idx = idx[temp]
    return lb[idx], ub[idx], idx


def _local_index(mapping, L):
    """global group index -> position inside one output (-1: the output does not have the group)"""
    local = np.full(L, -1, dtype=np.int64)
    local[np.asarray(mapping)] = np.arange(len(mapping))
    return local


def psi_column(sap, li):
    """column li of SAP_n's psi (cmisc.cpp:10-23) from the group's inverse covariance: (N*N,) vector"""
    N = sap.N
    k = int(np.searchsorted(sap.cumsizes, li, side="right"))          # group size
    i = li - int(sap.cumsizes[k - 1])
    g = sap.groups[k - 1][i]
    ic = np.asarray(sap.invcovs[k - 1][i * k * k:(i + 1) * k * k]).reshape(k, k)
    col = np.zeros(N * N)
    col[(N * g[:, None] + g[None, :]).ravel()] = ic.ravel()
    return col


def candidate_variances(N, base_phi, cols, ms, device):
    """GPU: V[c, o] = pinv(base_phi[o] + sum_j ms[j, c] cols[o, j])[0, 0]; ms is (LL, n_c) as in the reference"""
    n_out, LL = cols.shape[0], cols.shape[1]
    n_c = ms.shape[1]
    base_d = base_phi.to(device=device, dtype=torch.float64).contiguous()
    cols_d = torch.from_numpy(np.ascontiguousarray(cols)).to(device)
    ms_d = torch.from_numpy(np.ascontiguousarray(ms.T, dtype=np.float64)).to(device)
    V = torch.empty((n_c, n_out), dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        check(_lib.lib().bluest_intproj_eval(N, n_out, LL, base_d.data_ptr(), cols_d.data_ptr(), ms_d.data_ptr(), n_c, V.data_ptr(), _stream()))
    return V


def _search(sol, N, w, e, saps, mappings, plan, budget, eps, max_samples_info, lb, ub, idx, multi):
    """bluest/misc.py:228-311 (multi) / :313-382 (single), chunked over the 2^LL candidates"""
    ES, rhs = max_samples_info
    No = len(saps)
    LL = len(idx)
    val = np.round(sol).astype(int)
    baseval = val.copy()
    baseval[idx] = 0
    basecost = w @ baseval
    basees = [e[mappings[n]] @ baseval[mappings[n]] for n in range(No)]
    base_max = [ees @ baseval for ees in ES]
    if len(ES) > 0 and any(b > rr for b, rr in zip(base_max, rhs)):
        return None, np.inf
    if budget is not None and basecost > budget:
        return None, np.inf
    need_e = [n for n in range(No) if basees[n] < 1]
    if multi and len(need_e) == 0:
        return None, np.inf        # the reference returns here as well (misc.py:263, `es` stays empty)

    # base information matrices and the psi columns of the free groups, per output
    base_phi = plan.phi_matrix(baseval.astype(np.float64))[0].reshape(No, N * N)
    cols = np.zeros((No, LL, N * N))
    in_out = np.zeros((No, LL), dtype=bool)
    for n in range(No):
        local = _local_index(mappings[n], len(sol))
        for j, gidx in enumerate(idx):
            li = int(local[gidx])
            if li >= 0:
                cols[n, j] = psi_column(saps[n], li)
                in_out[n, j] = True

    bnds = np.vstack([lb, ub])
    best = (None, np.inf, -np.inf)   # (ms column, objective V_max, cost) under the reference's selection rule
    found_any = False
    total = 1 << LL
    step = 1 << min(LL, CHUNK_BITS)
    weights = 2 ** np.arange(LL, dtype=np.int64)
    for start in range(0, total, step):
        codes = np.arange(start, min(start + step, total), dtype=np.int64)
        combs = ((codes[:, None] & weights[None, :]) > 0).astype(np.int64)      # misc.py:169-175 unpackbits
        ms = bnds[combs, np.arange(LL)[None, :]].T                               # (LL, n_c)
        keep = np.ones(ms.shape[1], dtype=bool)
        if not multi:
            if basees[0] < 1:
                keep &= basees[0] + e[idx] @ ms >= 1                             # misc.py:337-342
        else:
            anyok = np.zeros(ms.shape[1], dtype=bool)                            # misc.py:257-265 (union over outputs)
            for n in need_e:
                anyok |= basees[n] + (e[idx] * in_out[n]) @ ms >= 1
            keep &= anyok
        for ees, b, rr in zip(ES, base_max, rhs):                               # misc.py:267-276, 344-353
            keep &= b + ees[idx] @ ms <= rr
        costs = basecost + w[idx] @ ms
        if budget is not None:
            keep &= costs <= 1.0001 * budget                                    # misc.py:284, 361
        if not keep.any():
            continue
        ms, costs = ms[:, keep], costs[keep]
        V = candidate_variances(N, base_phi, cols, ms, plan.device)
        Vmax = V.cpu().numpy().max(axis=1)
        if budget is not None:
            i = int(np.argmin(Vmax))
            # the reference reverses the candidate order before argmin: on exact ties the LAST candidate wins
            ties = np.nonzero(Vmax == Vmax[i])[0]
            i = int(ties[-1])
            if np.isfinite(Vmax[i]) and (Vmax[i] < best[1] or (Vmax[i] == best[1])):
                best = (ms[:, i].copy(), float(Vmax[i]), float(costs[i]))
                found_any = True
        else:
            Vh = V.cpu().numpy()
            ok = np.all(Vh <= 1.0001 * np.asarray(eps)[None, :] ** 2, axis=1)    # misc.py:301, 375
            if ok.any():
                cand = np.nonzero(ok)[0]
                i = int(cand[np.argmin(costs[cand])])                           # cheapest feasible (misc.py:289,302)
                if not found_any or costs[i] < best[2]:
                    best = (ms[:, i].copy(), float(Vmax[i]), float(costs[i]))
                found_any = True
    if not found_any:
        return None, np.inf
    val[idx] = best[0]
    return val, best[1]


def best_closest_integer_solution(sol, N, w, e, saps, mappings, plan, budget=None, eps=None, max_samples_info=([], []),
                                  LL_max=15, rng=None, multi=True):
    """bluest/misc.py:177-226 (multi) and :313-321 (single: LL <= 24 or ValueError)"""
    lb_full, ub_full, idx_full = get_feasible_integer_bounds(sol, N, e=e)
    LL = len(idx_full)
    if not multi and LL <= 24:
        return _search(sol, N, w, e, saps, mappings, plan, budget, eps, max_samples_info, lb_full, ub_full, idx_full, False)
    if not multi:
        # the reference raises ValueError('Too many dimensions to brute-force it') here (misc.py:320-321); with the candidate
        # batch on the GPU the randomised search of the multi-output version (misc.py:192-226) is affordable instead
        LL_max = 15
    if multi and LL <= LL_max:
        return _search(sol, N, w, e, saps, mappings, plan, budget, eps, max_samples_info, lb_full, ub_full, idx_full, True)
    # misc.py:192-226: brute-force a random subset of LL_max entries, randomise the rest, up to 250 trials
    print('WARNING! Too many dimensions to brute-force it. Randomising search. Note: result might not be optimal.')
    rng = np.random.RandomState(0) if rng is None else rng      # reproducible (the reference uses the global numpy state)
    best_val, best_fval, trial = None, np.inf, 0
    while best_val is None and trial < 250:
        trial += 1
        sample = rng.permutation(LL)
        brute, rest = sample[:LL_max], sample[LL_max:]
        r_sol = sol.copy()
        r_bnds = np.vstack([lb_full[rest], ub_full[rest]])
        comb = rng.randint(2, size=len(rest))
        r_sol[idx_full[rest]] = r_bnds[comb, np.arange(len(rest))]
        best_val, best_fval = _search(r_sol, N, w, e, saps, mappings, plan, budget, eps, max_samples_info,
                                      lb_full[brute], ub_full[brute], idx_full[brute], multi)
    # extension (no reference counterpart): with this many free entries the random search rarely hits the feasible corner,
    # so a greedy rounding competes with it; the better feasible point wins (lower variance / lower cost)
    g_val, g_fval = greedy_integer(sol, N, w, e, saps, mappings, plan, budget, eps, max_samples_info)
    if g_val is not None:
        if best_val is None:
            better = True
        elif budget is not None:
            better = g_fval < best_fval
        else:
            better = w @ g_val < w @ best_val
        if better:
            best_val, best_fval = g_val, g_fval
    if best_val is None:
        print("Unable to find feasible integer solution.")
        return None, np.inf
    return best_val, best_fval


def greedy_integer(sol, N, w, e, saps, mappings, plan, budget=None, eps=None, max_samples_info=([], []), max_support=256):
    """Greedy rounding of a continuous allocation on its support, every step one GPU batch of single-sample moves
    (`bluest_intproj_eval`).  budget mode: start from floor(sol), keep buying the sample with the largest drop of max_o V_o
    per unit cost that still fits 1.0001*budget (misc.py:284).  eps mode: start from ceil(sol), keep removing the most
    expensive sample that leaves every V_o <= 1.0001*eps_o^2 (misc.py:301).  Returns (integer allocation, max_o V_o) or
    (None, inf).  Not in the reference: used next to its randomised search when there are too many free entries."""
    ES, rhs = max_samples_info
    No = len(saps)
    sup = np.flatnonzero(sol > 1.0e-8)
    if len(sup) == 0:
        return None, np.inf
    if len(sup) > max_support:
        sup = sup[np.argsort(sol[sup])[-max_support:]]
    LL = len(sup)
    cols = np.zeros((No, LL, N * N))
    locals_ = [_local_index(mappings[n], len(sol)) for n in range(No)]
    for n in range(No):
        for j, gidx in enumerate(sup):
            li = int(locals_[n][gidx])
            if li >= 0:
                cols[n, j] = psi_column(saps[n], li)
    wsup = w[sup]

    def sampled_once(mv):
        return all(e[mappings[n]] @ mv[mappings[n]] >= 1 for n in range(No))

    def within_caps(mv):
        return all(ees @ mv <= rr for ees, rr in zip(ES, rhs))

    def moves(mv, sign):
        """V[j, o] of mv + sign * (one sample of support entry j); the candidate kernel takes <= 32 free entries per call"""
        base = plan.phi_matrix(mv.astype(np.float64))[0].reshape(No, N * N)
        out = []
        for c0 in range(0, LL, 32):
            c1 = min(c0 + 32, LL)
            out.append(candidate_variances(N, base, cols[:, c0:c1], sign * np.eye(c1 - c0), plan.device).cpu().numpy())
        V = np.vstack(out)
        return np.where(np.isfinite(V), V, np.inf)

    def variances(mv):
        base = plan.phi_matrix(mv.astype(np.float64))[0].reshape(No, N * N)
        V = candidate_variances(N, base, cols[:, :1], np.zeros((1, 1)), plan.device).cpu().numpy()[0]
        return np.where(np.isfinite(V), V, np.inf)

    m = np.zeros(len(sol), dtype=np.int64)
    if budget is not None:
        m[sup] = np.floor(sol[sup]).astype(np.int64)
        while not sampled_once(m):                         # model 0 of some output is not sampled yet: cheapest group that has it
            lacking = [n for n in range(No) if e[mappings[n]] @ m[mappings[n]] < 1]
            cand = [j for j in range(LL) if any(e[sup[j]] > 0 and locals_[n][sup[j]] >= 0 for n in lacking)]
            if not cand:
                return None, np.inf
            m[sup[min(cand, key=lambda j: wsup[j])]] += 1
        if w @ m > 1.0001 * budget or not within_caps(m):
            return None, np.inf
        Vcur = variances(m).max()
        for _ in range(3 * LL + 16):
            left = 1.0001 * budget - w @ m
            V = moves(m, +1).max(axis=1)
            gain = (Vcur - V) / wsup
            gain[wsup > left] = -np.inf
            for j in np.argsort(-gain):
                if not gain[j] > 0.0:
                    j = -1
                    break
                trial = m.copy()
                trial[sup[j]] += 1
                if within_caps(trial):
                    break
            else:
                j = -1
            if j < 0:
                break
            m[sup[j]] += 1
            Vcur = V[j]
        return (m, float(Vcur)) if np.isfinite(Vcur) else (None, np.inf)
    lim = 1.0001 * np.asarray(eps, dtype=np.float64) ** 2
    m[sup] = np.ceil(sol[sup]).astype(np.int64)
    Vcur = variances(m)
    for _ in range(3 * LL + 16):                          # the continuous point was infeasible after pruning: buy samples first
        if (Vcur <= lim).all():
            break
        V = moves(m, +1)
        j = int(np.argmin((V / lim[None, :]).max(axis=1) + 1e-12 * wsup))
        m[sup[j]] += 1
        Vcur = V[j]
    if not (Vcur <= lim).all() or not sampled_once(m) or not within_caps(m):
        return None, np.inf
    for _ in range(3 * LL + 16):
        V = moves(m, -1)
        ok = (V <= lim[None, :]).all(axis=1) & (m[sup] >= 1)
        best = -1
        for j in np.argsort(-wsup):
            if ok[j]:
                trial = m.copy()
                trial[sup[j]] -= 1
                if sampled_once(trial):
                    best = j
                    break
        if best < 0:
            break
        m[sup[best]] -= 1
        Vcur = V[best]
    return m, float(Vcur.max())


def _increase_tolerance(budget, eps, fac):
    b = None if budget is None else budget * (1 + fac)
    e = None if eps is None else np.sqrt(np.asarray(eps, dtype=np.float64) ** 2 * (1 + fac))
    return b, e


def integer_projection_sap(sap, samples, budget=None, eps=None, max_model_samples=None):
    """bluest/sap.py:145-187"""
    if budget is None and eps is None:
        raise ValueError("Need to specify either budget or RMSE tolerance")
    if sap.verbose: print("Integer projection...")
    ss = samples.copy()
    es, rhs = sap.get_max_sample_constraints(max_model_samples)
    maps = [np.arange(sap.L)]
    epsv = None if eps is None else [eps]
    args = (sap.N, sap.costs, sap.e, [sap], maps, sap.plan)
    out, fval = best_closest_integer_solution(ss, *args, budget=budget, eps=epsv, max_samples_info=(es, rhs), multi=False)
    if np.isinf(fval):
        for i in reversed(range(4)):
            if sap.verbose: print("WARNING! An integer solution satisfying the constraints could not be found. Increasing the tolerance/budget.\n")
            # NOTE: the reference computes the relaxed tolerances and then passes the ORIGINAL ones (sap.py:170-171);
            # kept as is, so this rung repeats STEP 0
            _increase_tolerance(budget, eps, 10. ** -i)
            out, fval = best_closest_integer_solution(ss, *args, budget=budget, eps=epsv, max_samples_info=(es, rhs), multi=False)
            if not np.isinf(fval): break
    if np.isinf(fval):
        if max_model_samples is not None and not all([np.ceil(ss) @ ee <= rr for ee, rr in zip(es, rhs)]):
            out = np.floor(ss)
            if not out @ sap.e >= 1.0:
                out = np.ceil(ss)
        else:
            if sap.verbose: print("WARNING! An integer solution satisfying the constraints could not be found even after increasing the tolerance/budget. Rounding up.\n")
            out = np.ceil(ss)
    return np.asarray(out).astype(int)


def integer_projection_mosap(mos, samples, budget=None, eps=None, max_model_samples=None):
    """bluest/mosap.py:212-289: the whole ladder -- closest integer point; the same from the cleaned-up (sparser) allocation;
    both again with budget / tolerances relaxed by 1e-3 ... 1; finally rounding up or down"""
    if budget is None and eps is None:
        raise ValueError("Need to specify either budget or RMSE tolerance")
    if mos.verbose: print("Integer projection...")
    ss = samples.copy()
    ES, rhs = mos.get_max_sample_constraints(max_model_samples)
    args = (mos.N, mos.costs, mos.e, mos.SAPS, mos.mappings, mos.plan)

    def closest(start, b, e):
        return best_closest_integer_solution(start, *args, budget=b, eps=e, max_samples_info=(ES, rhs))

    out, fval = closest(ss, budget, eps)                                   # STEP 0
    css = None
    if np.isinf(fval):                                                     # STEP 1: clean up, try again
        if mos.verbose: print("Integer projection failed. Trying to recover by cleanup...")
        css = mos.cleanup_solution(ss)
        out, fval = closest(css, budget, eps)
    if np.isinf(fval):                                                     # STEP 2: relax the constraints
        for i in reversed(range(4)):
            if mos.verbose: print("WARNING! An integer solution satisfying the constraints could not be found. Increasing the tolerance/budget.\n")
            nb, ne = _increase_tolerance(budget, eps, 10. ** -i)
            out, fval = closest(ss, nb, ne)
            if np.isinf(fval):
                out, fval = closest(css, nb, ne)
            if not np.isinf(fval): break
    if np.isinf(fval):                                                     # STEP 3: round
        def sampled_once(x):
            return all([x[mos.mappings[n]] @ mos.e[mos.mappings[n]] >= 1 for n in range(mos.n_outputs)])

        def within_caps(x):
            return all([x @ ees <= rr for ees, rr in zip(ES, rhs)])

        up, down, cup, cdown = np.ceil(ss), np.floor(ss), np.ceil(css), np.floor(css)
        if eps is None: prefer_up = up @ mos.costs < cup @ mos.costs
        else:           prefer_up = max(mos.variances(up)) < max(mos.variances(cup))
        warn = "WARNING! An integer solution satisfying the constraints could not be found even after increasing the tolerance/budget."
        if max_model_samples is not None and within_caps(up):
            out, note = up, " Rounding up.\n"
        elif max_model_samples is not None and within_caps(cup):
            out, note = cup, " Rounding up.\n"
        elif max_model_samples is not None and sampled_once(down):
            out, note = down, " Rounding down to satisfy max model sample constraints.\n"
        elif max_model_samples is not None and sampled_once(cdown):
            out, note = cdown, " Rounding down to satisfy max model sample constraints.\n"
        else:
            out = up if prefer_up else cup
            note = (" Rounding up.\n" if max_model_samples is None else
                    " and the max model sample constraints could not be satisfied. Rounding up.\n")
        if mos.verbose: print(warn + note)
    return np.asarray(out).astype(int)
