"""Where the host spends a warm MOSAP.solve: python tools/host_profile.py [n k n_out]   (cProfile of the third solve, and the
segment clock of colgen_solve: solver_params={"newton": {"profile": True}})"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.mosap import MOSAP  # noqa: E402

n, k, o = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (20, 5, 8)
prob = synth.problem(n, k, o)
mos = MOSAP(prob["C"], k, [k] * o, [g.copy() for g in prob["groups"]], [[g.copy() for g in prob["groups"]] for _ in range(o)],
            prob["costs"], [prob["costs"]] * o, verbose=False)
for rep in range(4):
    prof = cProfile.Profile() if rep == 3 else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    m = mos.solve(budget=prob["budget"], solver="spg", continuous_relaxation=True, solver_params={"newton": {"profile": rep == 2}})
    if prof:
        prof.disable()
    torch.cuda.synchronize()
    print("rep %d: solve %.2f ms  gap %.2e rounds %s" % (rep, (time.perf_counter() - t0) * 1e3, mos.solver_info.get("certified_gap", float("nan")),
                                                       mos.solver_info.get("rounds")), flush=True)
    if rep == 2:
        print("   colgen host segments (ms):", mos.solver_info.get("host_ms"))
    if prof:
        pstats.Stats(prof).sort_stats("cumulative").print_stats(40)
