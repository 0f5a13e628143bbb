"""certified gap of the ragged Navier-Stokes problem (eps mode) under small perturbations of the solver's parameters: how much the
end point of the second-order finish depends on the last bits of the trajectory"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402
from test_oracle import _ns_case  # noqa: E402

G = dict(np.load(os.path.join(ROOT, "tests", "golden", "ns_paper_known_answer.npz")))
n_out, kmax = int(G["n_out"]), int(G["kmax"])
for case in ("ragged", "full"):
    groups, maps, multi = _ns_case(G, case)
    Cs = [G["C%d" % o] for o in range(n_out)]
    costs = synth.group_costs(groups, G["costs"])
    for prm in ({}, {"ma_p": 31.0}, {"ma_p": 33.0}, {"ma_p": 28.0}, {"ma_iterations": 190}, {"ma_iterations": 210}, {"ma_iterations": 150},
                {"support_init": 4}, {"enter_per_round": 8}):
        mos = MOSAP(Cs, kmax, [kmax] * n_out, [g.tolist() for g in groups], [[g.tolist() for g in mg] for mg in multi], costs,
                    [synth.group_costs(mg, G["costs"]) for mg in multi], verbose=False)
        m = mos.solve(eps=list(G["eps"]), solver="spg", continuous_relaxation=True, solver_params={"newton": prm})
        si = mos.solver_info
        print(case, prm, "gap %.2e" % si.get("certified_gap", np.nan), "method", si.get("method"), "rounds", si.get("rounds"), "it", si.get("it"), "cost %.8g" % float(m @ costs))
