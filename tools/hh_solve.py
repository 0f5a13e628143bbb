"""End-to-end known answer on real data: Hodgkin-Huxley paper problem (n=12, n_out=5, K=7, K_tot=3301), eps mode.
The stored allocation of the paper (SDP solver + integer projection) costs 60626.8 with errors/eps <= 1.00004."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
G = dict(np.load("tests/golden/hh_paper_known_answer.npz"))
n, n_out, kmax = int(G["n"]), int(G["n_out"]), int(G["kmax"])
groups = synth.all_groups(n, kmax)
costs = synth.group_costs(groups, G["costs"])
Cs = [G["C%d" % o] for o in range(n_out)]
eps = np.sqrt(np.array([C[0, 0] for C in Cs])) / 1000
t0 = time.perf_counter()
mos = MOSAP(Cs, kmax, [kmax] * n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)], costs, [costs] * n_out, verbose=False)
t1 = time.perf_counter()
out = {"setup_s": t1 - t0, "paper_cost": float(G["total_cost"])}
for maxit in (1000, 2000):
    for p in (32.0, (32.0, 512.0), (32.0, 256.0, 4096.0)):
        t2 = time.perf_counter()
        mc = mos.solve(eps=eps, solver="spg", continuous_relaxation=True, solver_params={"maxit": maxit, "smoothing_p": p})
        t3 = time.perf_counter()
        V = np.array(mos.variances(mc))
        out["cont_maxit%d_p%s" % (maxit, p)] = {"cost": float(mc @ costs), "max_err_over_eps": float((np.sqrt(V) / eps).max()), "solve_s": t3 - t2,
                                        "info": {k: float(v) for k, v in mos.solver_info.items()}}
t2 = time.perf_counter()
mi = mos.solve(eps=eps, solver="spg", solver_params={"maxit": 2000})
t3 = time.perf_counter()
V = np.array(mos.variances(mi))
out["integer"] = {"cost": float(mi @ costs), "errs_over_eps": (np.sqrt(V) / eps).tolist(), "nnz": int((mi > 0).sum()), "solve_s": t3 - t2}
print(json.dumps(out, indent=1))
