set -x
mkdir -p gpurun_out
for N in 2 4; do
BLUEST_BENCH_BACKEND=gloo BLUEST_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 400 --warmup 40 > gpurun_out/r2_bench_share$N.json 2> gpurun_out/r2_bench_share$N.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r2_bench_share$N.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("n_gpus", "value", "ms_per_step", "exchange", "collective_backend", "ranks_seen", "single_gpu_value", "replica_value", "scaling")}, d["config"]["launch"])
PY
grep -iE "error|unavailable|Traceback" gpurun_out/r2_bench_share$N.err | head -5
done
BLUEST_BENCH_BACKEND=gloo BLUEST_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 400 --warmup 40 --no-graph > gpurun_out/r2_bench_share2_eager.json 2> /dev/null
python -c "
import json
d = json.loads(open('gpurun_out/r2_bench_share2_eager.json').read().strip().splitlines()[-1]); print('eager', d['ms_per_step'], d['config']['launch'])"
