import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from bluest_amd import synth
from bluest_amd.mosap import MOSAP
from bluest_amd.colgen import colgen_solve
G = dict(np.load("/root/repo/tests/golden/ns_paper_known_answer.npz"))
n, n_out, kmax = int(G["n"]), int(G["n_out"]), int(G["kmax"])
Cs = [G["C%d" % o] for o in range(n_out)]
groups = synth.all_groups(n, kmax)
costs = synth.group_costs(groups, G["costs"])
mos = MOSAP(Cs, kmax, [kmax]*n_out, [g.copy() for g in groups], [[g.copy() for g in groups] for _ in range(n_out)], costs, [costs]*n_out, verbose=False)
s = G["eps"] ** 2
B = 177375.97
x, info = colgen_solve(mos.plan, costs, s, B, log=print)
print({k: v for k, v in info.items()})
np.savez("/root/repo/gpurun_out/ns_dbg.npz", x=x, mu=info["mu"], B=B)
