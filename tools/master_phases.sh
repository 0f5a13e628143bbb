#!/bin/bash
# Per-phase time of k_master_newton over whole solves (experiment build -DMASTER_TIMING: thread 0's wall clock between workgroup
# barriers; the product build is restored on exit):  tools/master_phases.sh out.txt
out=${1:-gpurun_out/master_phases.txt}
trap 'env -u BLUEST_EXTRA_HIPCC_FLAGS python -m bluest_amd.build --force > /dev/null 2>&1' EXIT
BLUEST_EXTRA_HIPCC_FLAGS="-DMASTER_TIMING $MASTER_PHASE_FLAGS" python -m bluest_amd.build --force > /dev/null 2>&1 || { echo "timing build failed" > $out; exit 1; }
python - > $out <<'PY'
import re, subprocess, sys
names = ["load of the support (blocks, lists)", "Phi assembly of the evaluations", "elimination of the evaluations (V only, DPP)",
         "active set + derivatives (T of the active outputs, a, gradients)", "free set / step formation / bookkeeping",
         "Hessian of the Lagrangian", "elimination of the Newton system (8 wavefronts)", "K = E^T M^-1 E + the small KKT system"]
print("# k_master_newton per phase, microseconds per SOLVE (all master calls of one warm colgen_solve), thread 0's wall clock between")
print("# workgroup barriers, experiment build -DMASTER_TIMING (tools/master_phases.sh).  Round 3 (before): profiles/r04_master_phases_before.txt")
import os
for cfg in (os.environ.get("MASTER_PHASE_CFGS", "20 5 8,25 6 1,20 5 1").split(",")):
    out = subprocess.run([sys.executable, "tools/colgen_run.py"] + cfg.split(), capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("rep 2")][-1]
    ph = [float(x) for x in re.findall(r"np\.float64\(([-0-9.e+]+)\)", line.split("master_phase_us")[1].split("]")[0])]
    g = lambda k: int(re.search(r"'%s': (\d+)" % k, line).group(1))
    its, evals, solves, rounds = g("newton_it"), g("master_evals"), g("master_solves"), g("rounds")
    print("\n== n, k_max, n_out = %s: solve %s, %d rounds, %d Newton iterations, %d evaluations, %d factorisations; master total %.0f us"
          % (cfg, line.split("s ")[0].split(": ")[1] + "s", rounds, its, evals, solves, sum(ph)))
    per = [rounds + 1, evals, evals, its, solves, solves, solves, solves]
    unit = ["call", "evaluation", "evaluation", "iteration", "factorisation", "factorisation", "factorisation", "factorisation"]
    for i, (nm, v) in enumerate(zip(names, ph)):
        print("  %d  %-66s %9.1f us  %5.1f %%   %6.2f us per %s" % (i, nm, v, 100 * v / sum(ph), v / max(per[i], 1), unit[i]))
PY
cat $out
