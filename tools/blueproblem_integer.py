"""BLUEProblem.setup_solver() through the user-facing API with the DEFAULT integer projection (continuous_relaxation=False),
budget mode and eps mode, on the headline problem (n=20, n_out=8, K=5)."""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from bluest_amd import BLUEProblem, synth  # noqa: E402

n, kmax, n_out = (int(a) for a in (sys.argv[1:4] or (20, 5, 8)))
prob = synth.problem(n, kmax, n_out)
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
p = BLUEProblem(n, C=[c.copy() for c in prob["C"]], costs=prob["w"], n_outputs=n_out, verbose=False)
eps = [float(np.sqrt(c[0, 0]) / 30.0) for c in prob["C"]]
for kw in ({"budget": prob["budget"]}, {"eps": eps}):
    for rep in range(2):
        t0 = time.perf_counter()
        out = p.setup_solver(K=kmax, solver="spg", **kw)
        dt = time.perf_counter() - t0
    samples = np.asarray(out["samples"])
    print("%-7s setup_solver %.3f s: %d groups sampled, integer %s, total cost %.4f, max error %.6e%s" % (
        list(kw)[0], dt, len(out["models"]), samples.dtype.kind == "i", out["total_cost"], np.max(out["errors"]),
        "" if "eps" not in kw else ", max error/eps %.5f" % np.max(np.asarray(out["errors"]) / np.asarray(eps))))
