"""the ragged / full Navier-Stokes problem in eps mode once, with the rounds of the second-order finish printed:
    BLUEST_COLGEN_LOG=1 python tools/ns_one.py [ragged|full]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402
from test_oracle import _ns_case  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "ragged"
G = dict(np.load(os.path.join(ROOT, "tests", "golden", "ns_paper_known_answer.npz")))
n_out, kmax = int(G["n_out"]), int(G["kmax"])
groups, maps, multi = _ns_case(G, case)
Cs = [G["C%d" % o] for o in range(n_out)]
costs = synth.group_costs(groups, G["costs"])
mos = MOSAP(Cs, kmax, [kmax] * n_out, [g.tolist() for g in groups], [[g.tolist() for g in mg] for mg in multi], costs,
            [synth.group_costs(mg, G["costs"]) for mg in multi], verbose=False)
m = mos.solve(eps=list(G["eps"]), solver="spg", continuous_relaxation=True)
si = mos.solver_info
print(case, "gap %.2e" % si.get("certified_gap", np.nan), "method", si.get("method"), "rounds", si.get("rounds"), "it", si.get("it"), "cost %.8g" % float(m @ costs))
