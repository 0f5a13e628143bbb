set -x
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -v -x -p no:cacheprovider -W ignore > gpurun_out/r2_gputests_f.log 2>&1
grep -E "PASSED|FAILED|ERROR|passed|failed" gpurun_out/r2_gputests_f.log | tail -60
grep -E "^E " gpurun_out/r2_gputests_f.log | head -20
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_h.json 2> gpurun_out/r2_bench_h.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2_bench_h.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernels_us"])
s = d["sap_wallclock"]; print("cold", s["cold"]); print("warm", s["warm"]); print(s["warm_all_total_s"])
PY
