#!/bin/bash
# round 5: the kg_conv kernel tests with every ring tile forced, REPS times in a row (a start-up race in the ring showed up once
# in four full runs before its fix)
mkdir -p gpurun_out
: > gpurun_out/ring_stress.log
for i in $(seq 1 ${REPS:-3}); do
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "conv and not aggconv and ring" 2>&1 | grep -E "FAILED|passed|failed" | tee -a gpurun_out/ring_stress.log
done
