import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
from bluest_amd.colgen import colgen_solve
from test_oracle import _ns_case
G = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ns_paper_known_answer.npz")))
n_out, kmax = int(G["n_out"]), int(G["kmax"])
groups, maps, multi = _ns_case(G, "ragged")
Cs = [G["C%d" % o] for o in range(n_out)]
costs = synth.group_costs(groups, G["costs"])
mos = MOSAP(Cs, kmax, [kmax] * n_out, [g.tolist() for g in groups], [[g.tolist() for g in mg] for mg in multi], costs,
            [synth.group_costs(mg, G["costs"]) for mg in multi], verbose=False)
x, info = colgen_solve(mos.plan, costs, G["eps"] ** 2, 1.7e5, log=print)
print(x is None, info if x is None else {k: v for k, v in info.items() if k not in ("certificate",)})
