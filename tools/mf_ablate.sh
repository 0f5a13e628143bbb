#!/bin/bash
# parts of the matrix-free Phi pass switched off one at a time (timing only, wrong numbers): tools/mf_ablate.sh out.txt
out=${1:-gpurun_out/mf_ablate.txt}
: > $out
trap 'env -u BLUEST_EXTRA_HIPCC_FLAGS python -m bluest_amd.build --force > /dev/null 2>&1' EXIT
for v in 0 1 2 3 4; do
  BLUEST_EXTRA_HIPCC_FLAGS="-DMF_ABLATE=$v" python -m bluest_amd.build --force > /dev/null 2>&1 || { echo "build $v failed" >> $out; continue; }
  echo "[MF_ABLATE=$v] $(timeout -k 10 200 python tools/matfree_ab.py 25 6 1 20 5 8 2>/dev/null | cut -c1-175 | tr '\n' ' ')" >> $out
done
cat $out
