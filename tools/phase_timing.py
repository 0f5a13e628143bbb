"""Experiment build only (BLUEST_EXTRA_HIPCC_FLAGS=-DBLUEST_PHASE_TIMING python -m bluest_amd.build --force): shader-clock
stamps of the solving wavefront and of the first tile wavefront of three workgroups of k_solve_grad, one evaluation step."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from bluest_amd import synth  # noqa: E402
from bluest_amd.plan import Plan  # noqa: E402

n, kmax, n_out = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else bench.HEADLINE
prob = synth.problem(n, kmax, n_out)
plan = Plan(n, prob["K_tot"], bench.build_outputs(prob))
m = torch.from_numpy(prob["m"][0]).to(plan.device)
for _ in range(20):
    plan.eval(m)
torch.cuda.synchronize()
lib = plan.lib
a = (ctypes.c_longlong * 36)()
b = (ctypes.c_longlong * 24)()
lib.bluest_debug_phase_times.argtypes = [ctypes.c_void_p]
lib.bluest_debug_phase_times_tile.argtypes = [ctypes.c_void_p]
lib.bluest_debug_phase_times(a)
lib.bluest_debug_phase_times_tile(b)
names = {0: "start", 1: "pads", 2: "fold+barrier", 8: "masks", 9: "rows loaded", 5: "elim begin", 6: "elim end", 7: "x ready", 3: "solve published"}
for w, wg in enumerate(("first", "middle", "last")):
    base = a[w * 12 + 0]
    print("workgroup %-6s solver wave :" % wg, "  ".join("%s %+d" % (names[i], a[w * 12 + i] - base) for i in (1, 2, 8, 9, 5, 6, 7, 3)))
    tb = b[w * 8 + 0]
    print("                 tile wave   : start %+d vs solver;" % (tb - base),
          "  ".join("%s %+d" % (nm, b[w * 8 + i] - base) for i, nm in ((1, "after fold barrier"), (2, "loads issued"), (3, "data landed"), (4, "after barrier 2"), (5, "grad stored"), (6, "store drained"))))
print("(shader cycles relative to the solving wavefront's first stamp; 2.4 cycles per ns)")
