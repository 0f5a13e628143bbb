#!/bin/bash
# the whole -m gpu suite with an unbuffered log under gpurun_out/ (a silent command is killed after 7 minutes on the GPU box)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -v -p no:cacheprovider -W ignore > gpurun_out/gpu_suite.log 2>&1
grep -E "PASSED|FAILED|ERROR|passed|failed" gpurun_out/gpu_suite.log | tail -80
grep -E "^E " gpurun_out/gpu_suite.log | head -30
