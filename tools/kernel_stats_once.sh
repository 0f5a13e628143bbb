#!/bin/bash
# rocprofv3 kernel statistics of one tools/one_solve.py run (top kernels):  bash tools/kernel_stats_once.sh [n k n_out reps]
ARGS=${@:-20 5 8 3}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_once
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_once -o t -- python3 $GRAFT_REPO_ROOT/tools/one_solve.py $ARGS > /tmp/once.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/tr_once/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print("%-50s %6s calls %8.2f us avg %6s %%" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep rep /tmp/once.log
