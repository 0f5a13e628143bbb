#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_spg; mkdir -p gpurun_out/prof_spg
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_spg -- python tools/sap_wallclock.py 20 5 8 > gpurun_out/prof_spg/out.json 2> gpurun_out/prof_spg/err.txt
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_spg/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-60s %8s %10.2f us avg  %6.2f%%" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
