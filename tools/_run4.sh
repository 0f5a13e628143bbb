set -x
mkdir -p gpurun_out
timeout -k 10 120 ./tools/micro/solve_bench > gpurun_out/r2_solve_bench.txt 2>&1
cat gpurun_out/r2_solve_bench.txt
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "tutorial or dropped or hypothesis" 2>&1 | grep -vE "Warning|setattr|_float_to_str|^$" | tail -40 > gpurun_out/r2_gputests_b.log
cat gpurun_out/r2_gputests_b.log
