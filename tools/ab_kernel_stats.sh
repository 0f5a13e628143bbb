#!/bin/bash
# A/B of one environment switch on the SAME box: rocprofv3 kernel statistics of tools/one_solve.py with and without it.
#   bash tools/ab_kernel_stats.sh BLUEST_PROJ_NO_BRACKET [n k n_out]
VAR=$1; shift
ARGS=${@:-20 5 8 3}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/ab_$VAR.txt
rm -f $OUT
for v in default $VAR; do
  if [ $v != default ]; then export $VAR=1; fi
  rm -rf /tmp/tr_ab
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_ab -o t -- python3 $GRAFT_REPO_ROOT/tools/one_solve.py $ARGS > /tmp/ab_run.log 2>&1
  echo "== $v" >> $OUT
  python3 - >> $OUT <<PY
import csv, glob
f = glob.glob("/tmp/tr_ab/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-50s %6s calls %8.2f us avg %6s %%" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  grep rep /tmp/ab_run.log >> $OUT
done
cat $OUT
