#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=line -p no:cacheprovider -k "agg or rowsum or batchnorm" > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log; tail -6 gpurun_out/kernels.log
timeout 600 python tools/profile_ops.py > gpurun_out/profile_ops.log 2>&1; tail -75 gpurun_out/profile_ops.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/bench_graph.log 2>&1; tail -1 gpurun_out/bench_graph.log | cut -c1-260
