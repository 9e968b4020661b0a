#!/bin/bash
# round 5: the bf16-split LDS-staged form of kg_conv (KG_CONV_BS) - kernel tests on its tiles, then the timing table
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv and (bs0 or bs1 or bs2)" 2>&1 | tail -5 | tee gpurun_out/bs_tests.log
bash tools/gpu_bs_time2.sh
