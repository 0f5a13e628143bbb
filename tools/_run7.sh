set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -v -x -p no:cacheprovider -W ignore > gpurun_out/r2_gputests_e.log 2>&1
grep -E "PASSED|FAILED|ERROR|passed|failed" gpurun_out/r2_gputests_e.log | tail -40
grep -E "^E " gpurun_out/r2_gputests_e.log | head -20
BLUEST_DEBUG_TIMING=1 timeout -k 10 200 python tools/setup_noise.py > gpurun_out/r2_setup_noise2.txt 2>&1
grep -A28 "^rep 1" gpurun_out/r2_setup_noise2.txt | head -60
timeout -k 10 300 python bench.py --no-cpu-baseline --no-batched > gpurun_out/r2_bench_e.json 2> gpurun_out/r2_bench_e.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2_bench_e.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"]); print(json.dumps(d["sap_wallclock"])[:900])
PY
