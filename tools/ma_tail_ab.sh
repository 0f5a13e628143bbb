#!/bin/bash
# does the multiplicative-update tail of k_solve_grad (one more kernel argument and a branch per tile) cost the benchmark step anything?
# same box, product build against -DBLUEST_NO_MA_TAIL, three bench runs each (restores the product build on exit)
trap 'env -u BLUEST_EXTRA_HIPCC_FLAGS python -m bluest_amd.build --force > /dev/null 2>&1' EXIT
run() { for i in 1 2 3; do python bench.py --no-cpu-baseline --no-sap --no-batched 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_us']; print('$1', 'ms_per_step %.5f' % d['ms_per_step'], 'chain step %.3f us' % k['step_us'], 'k_solve_grad %.3f us' % k['k_solve_grad_us(step - chunks; fused solve+gradient)'])"; done; }
python -m bluest_amd.build --force > /dev/null 2>&1; run "with tail   "
BLUEST_EXTRA_HIPCC_FLAGS="-DBLUEST_NO_MA_TAIL" python -m bluest_amd.build --force > /dev/null 2>&1; run "without tail"
env -u BLUEST_EXTRA_HIPCC_FLAGS python -m bluest_amd.build --force > /dev/null 2>&1; run "with tail   "
