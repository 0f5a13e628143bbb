set -x
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -k "not certified" 2>&1 | tail -40 > gpurun_out/r2_gputests_a.log
tail -15 gpurun_out/r2_gputests_a.log
timeout -k 10 400 python bench.py > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err
tail -c 3000 gpurun_out/r2_bench_a.json; tail -5 gpurun_out/r2_bench_a.err
BLUEST_BENCH_BACKEND=gloo BLUEST_BENCH_SHARE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 200 --warmup 20 > gpurun_out/r2_bench_g2.json 2> gpurun_out/r2_bench_g2.err
tail -c 2500 gpurun_out/r2_bench_g2.json; tail -5 gpurun_out/r2_bench_g2.err
