#!/bin/bash
# the multiplicative phase's parameters against time, rounds and certified gap over a few shapes:  tools/ma_sweep.sh out.txt "k=v k=v" ...
out=$1; shift
: > $out
for v in "$@"; do
  for cfg in "20 5 8" "25 6 1" "20 5 1" "12 12 1" "16 4 3" "24 4 6"; do
    echo "[$v] $cfg | $(timeout -k 10 120 python tools/colgen_run.py $cfg $v 2>/dev/null | tail -1 | cut -c1-60,75-330)" >> $out
  done
done
cat $out
