#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -p no:cacheprovider -k "wgrad or big" > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log; tail -6 gpurun_out/kernels.log
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/parity.log; tail -3 gpurun_out/parity.log
for m in 1 0; do
  KG_PARAM_DEFER=$m timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/bench_defer$m.log 2>&1
  echo "defer=$m: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_defer$m.log) $(grep -o '"d_only_ms_per_step": [0-9.]*' gpurun_out/bench_defer$m.log)"
done
