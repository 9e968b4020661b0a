#!/bin/bash
# round 5: the kg_conv kernel tests with one ring tile forced (RT=6), all failures listed
mkdir -p gpurun_out
for t in ${RT:-6 7 8 9}; do
  echo "== ring$t"
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv and not aggconv and ring$t" 2>&1 | grep -E "FAILED|passed|failed|Error" | head -40
done 2>&1 | tee gpurun_out/ringtest.log
