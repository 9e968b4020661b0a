#!/bin/bash
# GPU box: the two GPU test files, summaries into gpurun_out/
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/suite_kernels.log
timeout 2400 python -m pytest tests -x -q -m gpu --ignore=tests/test_kernels_gpu.py 2>&1 | tail -4 | tee gpurun_out/suite_rest.log
