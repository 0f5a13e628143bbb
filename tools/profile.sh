#!/bin/bash
# Round profile: (1) rocprofv3 --kernel-trace --stats of the default bench command, (2) separate --pmc passes for
# FETCH_SIZE and WRITE_SIZE (guide: MI355X_MICROARCH.md, HBM section).  Run on the GPU box:  bash tools/profile.sh r01
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python bench.py --no-cpu-baseline --no-graph --steps 200 --warmup 20 > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python bench.py --no-cpu-baseline --no-graph --steps 200 --warmup 20 > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err
python bench.py > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name "*.csv" | head -20
tail -2 $OUT/pmc_fetch.err
