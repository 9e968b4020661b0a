#!/bin/bash
# GPU visit: kernel tests + agg_outer timing
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=line -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log; tail -12 gpurun_out/kernels.log
timeout 300 python tools/time_wgrad.py > gpurun_out/time_wgrad.log 2>&1; cat gpurun_out/time_wgrad.log
