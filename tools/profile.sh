#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the bench command (without the SAP / CPU / batched legs, so the trace
# holds the timed hot path), (2) separate --pmc passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md, HBM section), (3) the
# default bench, (4) the other BASELINE configurations (n=20 single output with its own kernel trace, n=12 all groups, n=25),
# (5) kernel statistics of a whole SAP solve; summaries go to gpurun_out/profiles_<tag>/ (raw per-dispatch CSVs stay behind).
#   on the GPU box:  bash tools/profile.sh r02
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
SUM=gpurun_out/profiles_$TAG
rm -rf $OUT $SUM; mkdir -p $OUT $SUM
LEAN="--no-cpu-baseline --no-sap --no-batched"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py $LEAN > $OUT/bench_under_trace.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python bench.py $LEAN --no-graph --steps 200 --warmup 20 > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python bench.py $LEAN --no-graph --steps 200 --warmup 20 > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err
echo "pmc write done"
python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
python tools/pmc_traffic.py $OUT $TAG $SUM > $SUM/summary.txt
cp "$(ls -t $OUT/trace/*/*_kernel_stats.csv | head -1)" $SUM/${TAG}_kernel_stats_full.csv
cp $OUT/bench.json $SUM/${TAG}_bench.json
cp $OUT/bench_under_trace.json $SUM/${TAG}_bench_under_trace.json
# BASELINE.json configs[2]: n=20, k_max=5, ONE output (non-shared Phi kernel), its own kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_o1 -- python bench.py $LEAN --n 20 --kmax 5 --n-out 1 > $OUT/bench_o1_under_trace.json 2> $OUT/trace_o1.err
python - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/trace_o1/*/*_kernel_stats.csv"))[-1]
with open("$SUM/${TAG}_kernel_stats_n20_k5_o1.csv", "w") as out:
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in csv.DictReader(open(f)):
        w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
PY
python bench.py --n 20 --kmax 5 --n-out 1 > $SUM/${TAG}_bench_n20_k5_o1.json 2> $OUT/bench_o1.err
echo "n20 o1 done"
python bench.py --n 12 --kmax 12 --n-out 1 > $SUM/${TAG}_bench_n12_k12_o1.json 2> $OUT/bench_n12.err
python bench.py --n 25 --kmax 6 --n-out 1 > $SUM/${TAG}_bench_n25_k6_o1.json 2> $OUT/bench_n25.err
echo "other configs done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_spg -- python tools/sap_wallclock.py 20 5 8 > $OUT/spg_out.json 2> $OUT/spg_err.txt
python - > $SUM/${TAG}_spg_loop_kernel_stats.txt <<PY
import csv, glob
f = sorted(glob.glob("$OUT/prof_spg/*/*_kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:18]:
    print("%-60s %8s %10.2f us avg  %6.2f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -3 $OUT/spg_out.json >> $SUM/${TAG}_spg_loop_kernel_stats.txt
# (6) average kernel timeline of one SPG step (full problem: anchor k_proj_fused; working set: anchor k_simplex)
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_step -o t -- python tools/one_solve.py 20 5 8 2 > $OUT/one_solve.log 2> $OUT/one_solve.err
F=$(find $OUT/trace_step -name "*kernel_trace.csv" | head -1)
{ echo "# full-problem steps (K_tot = 21699)"; python tools/iter_timeline.py $F k_proj_fused; echo; echo "# working-set steps"; python tools/iter_timeline.py $F k_simplex; grep rep $OUT/one_solve.log; } > $SUM/${TAG}_spg_step_timeline.txt 2>&1
echo "step timeline done"
rm -rf $OUT
ls -la $SUM
