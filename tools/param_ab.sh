#!/bin/bash
# total solve time / iterations / objectives over tools/shape_times.py's shapes for several solver_params (JSON), same box
i=0
for sp in "$@"; do
  i=$((i+1))
  SOLVER_PARAMS="$sp" python tools/shape_times.py 2>&1 | grep -E "^n=|TOTAL" > gpurun_out/param_ab_$i.log
  echo "== $sp"; tail -1 gpurun_out/param_ab_$i.log
done
python - <<PY
import glob, re
files = sorted(glob.glob("gpurun_out/param_ab_*.log"))
rows = [[l for l in open(f) if l.startswith("n=")] for f in files]
best = [min(float(re.search(r"V ([0-9.e+-]+)", rows[j][i]).group(1)) for j in range(len(files))) for i in range(len(rows[0]))]
for j, f in enumerate(files):
    gaps = [float(re.search(r"V ([0-9.e+-]+)", rows[j][i]).group(1)) / best[i] - 1 for i in range(len(best))]
    print(f, "worst gap vs best of the variants %.1e, mean %.1e" % (max(gaps), sum(gaps) / len(gaps)))
PY
