#!/bin/bash
# Experiment: chain-clock step time of experiment builds (extra hipcc flags per variant, e.g. -DBLUEST_ABLATE=n: parts of the
# evaluation kernels switched off, see plan.hip).  Rebuilds the library on the GPU box per variant and restores the product
# build at the end.  Timings only: the ablated builds compute wrong numbers.
#   usage: tools/ablate.sh out.txt "flags of variant 1" "flags of variant 2" ...
out=$1; shift
: > $out
trap 'env -u BLUEST_EXTRA_HIPCC_FLAGS python -m bluest_amd.build --force > /dev/null 2>&1' EXIT     # product build back, whatever ends the script
for v in "$@"; do
  BLUEST_EXTRA_HIPCC_FLAGS="$v" python -m bluest_amd.build --force > /dev/null 2>&1 || { echo "build '$v' failed" >> $out; continue; }
  echo "[$v] $(timeout -k 10 120 python tools/step_parts.py $ABLATE_ARGS 2>/dev/null | tail -1)" >> $out
done
cat $out
