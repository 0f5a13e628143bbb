set -x
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-cpu-baseline --no-sap --no-batched > gpurun_out/r2_bench_i.json 2> gpurun_out/r2_bench_i.err
BLUEST_NO_STAGED_FOLD=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-sap --no-batched > gpurun_out/r2_bench_j.json 2> gpurun_out/r2_bench_j.err
python - <<'PY'
import json
for f in ("i", "j"):
    d = json.load(open("gpurun_out/r2_bench_%s.json" % f))
    print(f, d["value"], d["ms_per_step"], d["kernels_us"]["step_us"], d["kernels_us"]["k_phi_chunks_us"], d["kernels_us"]["separate_path"])
PY
timeout -k 10 600 python -m pytest tests -m gpu -v -x -p no:cacheprovider -W ignore -k "max_model or candidate_batch or properties_n20 or golden_n20" > gpurun_out/r2_gputests_g.log 2>&1
grep -E "PASSED|FAILED|ERROR|passed|failed" gpurun_out/r2_gputests_g.log | tail; grep -E "^E " gpurun_out/r2_gputests_g.log | head
