#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the bench command (without the SAP / CPU / batched legs, so the trace
# holds the timed hot path), (2) separate --pmc passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md, HBM section), (3) the
# default bench, (4) the other BASELINE configurations (n=20 single output with its own kernel trace, n=12 all groups, n=25),
# (5) kernel statistics of whole SAP solves, (6) the driver's command line, (7) step parts + quality sweep; summaries go to gpurun_out/profiles_<tag>/ (raw per-dispatch CSVs stay behind).
#   on the GPU box:  bash tools/profile.sh r02
TAG=${1:-r03}
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
# (5) kernel statistics of whole SAP solves (set-up + second-order finish), headline and the other BASELINE sizes
for cfg in "20 5 8" "20 5 1" "25 6 1"; do
  tagc=$(echo $cfg | tr " " _)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_solve_$tagc -- python tools/one_solve.py $cfg 3 > $OUT/solve_$tagc.log 2> $OUT/solve_$tagc.err
  python - > $SUM/${TAG}_solve_kernel_stats_$tagc.txt <<PY
import csv, glob
f = sorted(glob.glob("$OUT/prof_solve_$tagc/*/*_kernel_stats.csv"))[-1]
print("# rocprofv3 --kernel-trace --stats of tools/one_solve.py $cfg 3 (three set-ups + solves); per-kernel totals over the run")
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-60s %8s calls %10.2f us avg %9.2f ms total %6.2f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
  grep rep $OUT/solve_$tagc.log >> $SUM/${TAG}_solve_kernel_stats_$tagc.txt
done
echo "solve stats done"
# (6) the driver's command line, verbatim
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $SUM/${TAG}_bench_driver_style.json 2> $OUT/bench_driver.err
# (7) parts of a step on the chain clock, quality sweep
python tools/step_parts.py > $SUM/${TAG}_step_parts.txt 2>/dev/null
python tools/quality_sweep.py > $SUM/${TAG}_quality_sweep.txt 2>/dev/null
echo "driver-style bench + sweeps done"
rm -rf $OUT
ls -la $SUM
