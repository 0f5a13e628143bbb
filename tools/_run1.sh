set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "n20_k5_single or sharded_hip or certified or dropped_during" 2>&1 | tail -40 > gpurun_out/r2_newtests.log
cat gpurun_out/r2_newtests.log | tail -30
timeout -k 10 120 python tools/teardown_timing.py > gpurun_out/r2_teardown.txt 2>&1
timeout -k 10 120 python tools/teardown_timing.py --explicit > gpurun_out/r2_teardown_explicit.txt 2>&1
cat gpurun_out/r2_teardown.txt gpurun_out/r2_teardown_explicit.txt
