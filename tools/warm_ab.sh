#!/bin/bash
# A/B of compile-time variants of the master kernel over a few shapes: tools/warm_ab.sh out.txt "flags A" "flags B" ...
out=$1; shift
: > $out
trap 'env -u BLUEST_EXTRA_HIPCC_FLAGS python -m bluest_amd.build --force > /dev/null 2>&1' EXIT
for v in "$@"; do
  BLUEST_EXTRA_HIPCC_FLAGS="$v" python -m bluest_amd.build --force > /dev/null 2>&1 || { echo "build '$v' failed" >> $out; continue; }
  for cfg in "20 5 8" "25 6 1" "20 5 1" "12 12 1" "16 4 3" "30 3 4"; do
    echo "[$v] $cfg | $(timeout -k 10 120 python tools/colgen_run.py $cfg $AB_ARGS 2>/dev/null | tail -1 | cut -c1-330)" >> $out
  done
done
cat $out
