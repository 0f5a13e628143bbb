#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the bench command (without the SAP / CPU legs, so the trace holds
# the timed hot path), (2) separate --pmc passes for FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md, HBM section), (3) the
# default bench, (4) summaries written to gpurun_out/profiles_<tag>/ (the raw per-dispatch CSVs are too large to bring back).
#   on the GPU box:  bash tools/profile.sh r01
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
SUM=gpurun_out/profiles_$TAG
rm -rf $OUT $SUM; mkdir -p $OUT $SUM
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --no-cpu-baseline --no-sap > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python bench.py --no-cpu-baseline --no-sap --no-graph --steps 200 --warmup 20 > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python bench.py --no-cpu-baseline --no-sap --no-graph --steps 200 --warmup 20 > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python tools/pmc_traffic.py $OUT $TAG $SUM > $SUM/summary.txt
cp "$(ls -t $OUT/trace/*/*_kernel_stats.csv | head -1)" $SUM/${TAG}_kernel_stats_full.csv
cp $OUT/bench.json $SUM/${TAG}_bench.json
cp $OUT/bench_under_trace.json $SUM/${TAG}_bench_under_trace.json
rm -rf $OUT
ls -la $SUM
