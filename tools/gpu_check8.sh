#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=line -p no:cacheprovider -k "wgrad or big" > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log; tail -4 gpurun_out/kernels.log
timeout 600 python tools/time_wgrad.py 2>&1 | cut -c1-100 > gpurun_out/time_wgrad.log; cat gpurun_out/time_wgrad.log
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/parity.log; tail -3 gpurun_out/parity.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/bench_graph.log 2>&1; tail -1 gpurun_out/bench_graph.log | cut -c1-260
