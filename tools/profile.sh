#!/bin/bash
# Round profile.  For EVERY BASELINE configuration (headline n=20/k=5/8 outputs, n=20/k=5/1, n=12 all groups, n=25/k=6):
#   (1) rocprofv3 --kernel-trace --stats of the bench command (without the SAP / CPU / batched legs, so the trace holds the timed
#       hot path), (2) separate --pmc passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md, HBM section; the program directly
#       after `--`), summarised by tools/pmc_traffic.py into <tag>_kernel_stats<suffix>.csv / <tag>_pmc_traffic<suffix>.json;
# then (3) the default bench of every configuration (its line reads the summaries of (1)-(2): run profile.sh, copy the summaries to
# profiles/, run it again for lines with frac_rocprof / traffic of the SAME build -- or accept the previous round's), (4) kernel
# statistics of whole SAP solves, (5) the driver's command line, (6) master phases, step parts, quality sweep.
# Summaries go to gpurun_out/profiles_<tag>/ (raw per-dispatch CSVs stay behind).     on the GPU box:  bash tools/profile.sh r04
TAG=${1:-r04}
STAGE=${2:-ABC}        # A: traces + PMC, B: bench lines, C: solve statistics, master phases, sweeps (a gpurun call is limited to 20 minutes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
SUM=gpurun_out/profiles_$TAG
rm -rf $OUT; mkdir -p $OUT $SUM
LEAN="--no-cpu-baseline --no-sap --no-batched"
if [[ "$STAGE" == *A* ]]; then
# (n=25 evaluates matrix-free by default; "25 6 1 stored" profiles the stored-inverse path of the same configuration: BLUEST_MATFREE=0)
for cfg in "20 5 8" "20 5 1" "12 12 1" "25 6 1" "25 6 1 stored"; do
  set -- $cfg
  if [ "$cfg" = "20 5 8" ]; then SFX=""; else SFX="_n$1_k$2_o$3"; fi
  if [ "$4" = "stored" ]; then export BLUEST_MATFREE=0; SFX="_stored${SFX}"; else unset BLUEST_MATFREE; fi
  ARGS="--n $1 --kmax $2 --n-out $3"
  D=$OUT/cfg$SFX; mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python bench.py $LEAN $ARGS > $D/bench_under_trace.json 2> $D/trace.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/pmc_fetch -- python bench.py $LEAN $ARGS --no-graph --steps 200 --warmup 20 > $D/bench_under_pmc_fetch.json 2> $D/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/pmc_write -- python bench.py $LEAN $ARGS --no-graph --steps 200 --warmup 20 > $D/bench_under_pmc_write.json 2> $D/pmc_write.err
  python tools/pmc_traffic.py $D $TAG $SUM "$SFX" > $SUM/summary$SFX.txt
  if [ -z "$SFX" ]; then
    cp "$(ls -t $D/trace/*/*_kernel_stats.csv | head -1)" $SUM/${TAG}_kernel_stats_full.csv
    cp $D/bench_under_trace.json $SUM/${TAG}_bench_under_trace.json
  fi
  echo "trace + pmc of $cfg done"
done
unset BLUEST_MATFREE
fi
if [[ "$STAGE" == *B* ]]; then
# (3) default bench lines
python bench.py > $SUM/${TAG}_bench.json 2> $OUT/bench.err
python bench.py --n 20 --kmax 5 --n-out 1 > $SUM/${TAG}_bench_n20_k5_o1.json 2> $OUT/bench_o1.err
python bench.py --n 12 --kmax 12 --n-out 1 > $SUM/${TAG}_bench_n12_k12_o1.json 2> $OUT/bench_n12.err
python bench.py --n 25 --kmax 6 --n-out 1 > $SUM/${TAG}_bench_n25_k6_o1.json 2> $OUT/bench_n25.err
# (5) the driver's command line, verbatim
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $SUM/${TAG}_bench_driver_style.json 2> $OUT/bench_driver.err
echo "bench lines done"
fi
if [[ "$STAGE" == *C* ]]; then
# (4) kernel statistics of whole SAP solves (set-up + second-order finish), headline and the other BASELINE sizes
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
# (6) master phases (experiment build, product build restored), parts of a step on the chain clock, quality sweep
bash tools/master_phases.sh $SUM/${TAG}_master_phases.txt > /dev/null 2>&1
python tools/step_parts.py > $SUM/${TAG}_step_parts.txt 2>/dev/null
python tools/matfree_ab.py 25 6 1 20 5 8 20 5 1 > $SUM/${TAG}_matfree_ab_final.txt 2>/dev/null
python tools/batch_bench.py > $SUM/${TAG}_batch_bench.txt 2>/dev/null; python tools/batch_bench.py 25 6 1 >> $SUM/${TAG}_batch_bench.txt 2>/dev/null; python tools/batch_bench.py 20 5 1 >> $SUM/${TAG}_batch_bench.txt 2>/dev/null
python tools/quality_sweep.py > $SUM/${TAG}_quality_sweep.txt 2>/dev/null
echo "sweeps done"
fi
rm -rf $OUT
ls -la $SUM
