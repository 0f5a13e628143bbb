set -x
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | grep -vE "Warning|setattr|_float_to_str|^$" | tail -25 > gpurun_out/r2_gputests_c.log
cat gpurun_out/r2_gputests_c.log
timeout -k 10 300 python bench.py --no-sap --no-cpu-baseline --no-batched > gpurun_out/r2_bench_c.json 2> gpurun_out/r2_bench_c.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2_bench_c.json"))
print(d["value"], d["ms_per_step"], d["kernels_us"], d["roofline"]["frac"])
PY
