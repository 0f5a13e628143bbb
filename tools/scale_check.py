"""largest configuration end to end (n=25, k_max=6, K_tot=245505): set-up, SPG, integer projection, through the operator API"""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from bluest_amd import synth
from bluest_amd.sap import SAP
n, kmax = 25, 6
prob = synth.problem(n, kmax, 1)
t0 = time.perf_counter()
sap = SAP(prob["C"][0], kmax, [g for g in prob["groups"]], prob["costs"], verbose=False)
t1 = time.perf_counter()
mc = sap.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True)
t2 = time.perf_counter()
mi = sap.solve(budget=prob["budget"], solver="spg")
t3 = time.perf_counter()
print("K_tot", sap.L, "setup %.3f s, continuous solve %.3f s (%s), integer solve %.3f s" % (t1 - t0, t2 - t1, sap.solver_info, t3 - t2))
print("V cont %.6e  V int %.6e  cost int %.4f (budget %.1f)  nnz int %d  MC variance %.6e" % (
    sap.variance(mc), sap.variance(mi.astype(float)), mi @ prob["costs"], prob["budget"], (mi > 0).sum(), prob["C"][0][0, 0] / 1000))
eps = np.sqrt(sap.variance(mc)) * 0.7
me = sap.solve(eps=eps, solver="spg")
print("eps mode: V/eps^2 %.6f cost %.3f nnz %d" % (sap.variance(me.astype(float)) / eps ** 2, me @ prob["costs"], (me > 0).sum()))
