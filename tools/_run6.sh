set -x
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -v -x -p no:cacheprovider -W ignore > gpurun_out/r2_gputests_d.log 2>&1
grep -E "PASSED|FAILED|ERROR|passed|failed" gpurun_out/r2_gputests_d.log | tail -80
grep -E "^E " gpurun_out/r2_gputests_d.log | head -30
