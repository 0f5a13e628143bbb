#!/bin/bash
# the master's damping schedule (x MASTER_DAMP_DOWN after a good step, x MASTER_DAMP_UP after a rejected one; shipped 0.1 / 10) against
# time, iterations and certified gap (tools/gap_table.py); experiment builds, the product build is restored on exit:  tools/damp_ab.sh "0.3 10" "0.3 4" ...
trap 'env -u BLUEST_EXTRA_HIPCC_FLAGS python -m bluest_amd.build --force > /dev/null 2>&1' EXIT
for v in "$@"; do
  set -- $v
  BLUEST_EXTRA_HIPCC_FLAGS="-DMASTER_DAMP_DOWN=$1 -DMASTER_DAMP_UP=$2 -DMASTER_DAMP_HOLD=${3:-0}" python -m bluest_amd.build --force > /dev/null 2>&1 || { echo "build failed for $v"; continue; }
  echo "== down x$1, up x$2, hold ${3:-0}"
  python tools/gap_table.py 2>&1 | grep -v "amdgpu\|^#" | cut -c1-64
done
