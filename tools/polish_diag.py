"""Diagnosis: is the working-set point optimal ON ITS SUPPORT?  SLSQP (epigraph form, GPU evaluations) restricted to the support
of the solver's answer, and to the union with the support of the better point found with polish_full_loose=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.optimize import minimize
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
n, kmax, n_out = 16, 5, 2
prob = synth.problem(n, kmax, n_out)
g16 = prob["groups"]
mos = MOSAP(prob["C"], kmax, [kmax] * n_out, [g.copy() for g in g16], [[g.copy() for g in g16] for _ in range(n_out)],
            prob["costs"], [prob["costs"]] * n_out, verbose=False)
B, costs = prob["budget"], prob["costs"]
m_a = mos.solve(budget=B, solver="spg", continuous_relaxation=True)
m_b = mos.solve(budget=B, solver="spg", continuous_relaxation=True, solver_params={"polish_full_loose": 1.0})
print("default", max(mos.variances(m_a)), "nnz", (m_a > 0).sum(), " loose=1", max(mos.variances(m_b)), "nnz", (m_b > 0).sum())
print("support overlap", ((m_a > 0) & (m_b > 0)).sum())


def restricted_opt(sup, m0):
    idx = np.flatnonzero(sup)
    V0 = max(mos.variances(m0))

    def full(z):
        m = np.zeros(mos.L)
        m[idx] = np.maximum(z[:-1], 0.0) * B / costs[idx]
        return m

    def cons(z):
        return z[-1] - np.array(mos.variances(full(z))) / V0

    def cons_jac(z):
        _, grads, _ = mos.variance_GH(full(z), nohess=True)
        J = np.zeros((n_out, len(z)))
        for o in range(n_out):
            g = np.zeros(mos.L)
            g[mos.mappings[o]] = grads[o]
            J[o, :-1] = -g[idx] * B / costs[idx] / V0
        J[:, -1] = 1.0
        return J
    x0 = m0[idx] * costs[idx] / B
    x0 = np.maximum(x0, 1e-9); x0 /= x0.sum()
    z0 = np.concatenate([x0, [1.0]])
    res = minimize(lambda z: z[-1], z0, jac=lambda z: np.eye(len(z))[-1], method="SLSQP",
                   constraints=[{"type": "ineq", "fun": cons, "jac": cons_jac},
                                {"type": "eq", "fun": lambda z: z[:-1].sum() - 1.0, "jac": lambda z: np.concatenate([np.ones(len(z) - 1), [0.0]])}],
                   bounds=[(0.0, 1.0)] * len(x0) + [(0.0, 10.0)], options={"maxiter": 300, "ftol": 1e-12})
    return res.x[-1] * V0, res.status, res.nit


t0 = time.time()
print("restricted optimum on the default answer's support:", restricted_opt(m_a > 0, m_a), "%.1f s" % (time.time() - t0))
print("restricted optimum on the union of both supports:  ", restricted_opt((m_a > 0) | (m_b > 0), m_a))
