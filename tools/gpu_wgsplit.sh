#!/bin/bash
# round 5: the bf16-split tiles of kg_wgrad (KG_WGRAD_SPLIT) - the kernel tests on them, then the critic step's 16 layers timed on both forms
mkdir -p gpurun_out
KG_WGRAD_SPLIT=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wgrad" 2>&1 | tail -15 | tee gpurun_out/wgsplit_tests.log
for s in 0 1; do
  echo "== KG_WGRAD_SPLIT=$s" | tee -a gpurun_out/wgsplit_time.log
  KG_WGRAD_SPLIT=$s timeout 600 python tools/time_wgrad_many.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/wgsplit_time.log
done
