"""
Synthetic inputs for tests and bench.py.  numpy only.

Covariance per output o: Wishart(nu=2n, Sigma)/nu with Sigma_ij = 0.9^|i-j|, seed 1234+o; model costs
w_i = 2^-i; groups = all k-subsets of range(n), k = 1..k_max, size-major lexicographic (equals the sorted
clique order of bluest/blue_models.py:500-501 for a complete model graph); group cost = sum of model costs
(bluest/blue_models.py:137-140); throughput vector m = 10*rand(K_tot) drawn from the same stream after Z.
"""
from itertools import combinations
from math import comb

import numpy as np


def wishart_covariance(n, o=0, rho=0.9):
    """returns (C, rng): C = (L Z)(L Z)^T / nu, nu = 2n, L = chol(Sigma); rng is positioned after Z"""
    rng = np.random.RandomState(1234 + o)
    nu = 2 * n
    Z = rng.randn(n, nu)
    idx = np.arange(n)
    Sigma = rho ** np.abs(idx[:, None] - idx[None, :])
    A = np.linalg.cholesky(Sigma) @ Z
    return (A @ A.T) / nu, rng


def all_groups(n, kmax):
    """list over k=1..kmax of int64 arrays (L_k, k), lexicographic"""
    return [np.array(list(combinations(range(n), k)), dtype=np.int64).reshape((-1, k)) for k in range(1, kmax + 1)]


def n_groups(n, kmax):
    return sum(comb(n, k) for k in range(1, kmax + 1))


def model_costs(n):
    return 2.0 ** (-np.arange(n))


def group_costs(groups, w):
    return np.concatenate([w[g].sum(axis=1) for g in groups])


def problem(n, kmax, n_out=1):
    """dict with C (list over outputs), groups, costs (group costs), w (model costs), m (list over outputs), budget"""
    Cs, ms = [], []
    K_tot = n_groups(n, kmax)
    for o in range(n_out):
        C, rng = wishart_covariance(n, o)
        Cs.append(C)
        ms.append(10.0 * rng.rand(K_tot))
    groups = all_groups(n, kmax)
    w = model_costs(n)
    return {"n": n, "kmax": kmax, "n_out": n_out, "K_tot": K_tot, "C": Cs, "groups": groups, "w": w,
            "costs": group_costs(groups, w), "m": ms, "budget": 1000.0 * w[0]}


def algorithmic_bytes(n, kmax):
    """canonical bytes per evaluation (Phi pass + grad pass) in the reference layout, SURVEY.md 8(d):
    B = 2*(8*sum L_k k^2 + 8*sum L_k k + 8*K_tot) + 8n"""
    nnz = sum(comb(n, k) * k * k for k in range(1, kmax + 1))
    nidx = sum(comb(n, k) * k for k in range(1, kmax + 1))
    K_tot = n_groups(n, kmax)
    b_phi = 8 * nnz + 8 * nidx + 8 * K_tot
    return {"phi": b_phi, "grad": b_phi + 8 * n, "eval": 2 * b_phi + 8 * n}
