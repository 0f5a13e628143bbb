set -x
mkdir -p gpurun_out
BLUEST_DEBUG_TIMING=1 timeout -k 10 200 python tools/setup_noise.py > gpurun_out/r2_setup_noise.txt 2>&1
tail -60 gpurun_out/r2_setup_noise.txt
timeout -k 10 900 python tools/price_matrix.py > gpurun_out/r2_price_matrix.txt 2>&1
cat gpurun_out/r2_price_matrix.txt
