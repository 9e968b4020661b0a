#!/bin/bash
# GPU visit: kernel tests + conv tile tuning
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=line -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log; tail -25 gpurun_out/kernels.log
timeout 900 python tools/tune_conv.py > gpurun_out/tune_conv.log 2>&1; cat gpurun_out/tune_conv.log
