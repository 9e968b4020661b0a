#!/bin/bash
# round 5: the hand-scheduled all-window instantiation of the bf16-split form (kg_conv_bsw_kernel): tests, timing against KG_CONV_BS_ASM=0, stamps
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv and (bs0 or bs1 or bs2)" 2>&1 | tail -5 | tee gpurun_out/bsw_tests.log
OUT=gpurun_out/bsw_time.log
: > $OUT
KG_EXP_TAG=direct timeout 300 python tools/exp_conv.py 2>&1 | grep RES | grep -v total >> $OUT
KG_CONV_BS=1 KG_CONV_BS_ASM=0 KG_EXP_TAG=bs timeout 300 python tools/exp_conv.py 2>&1 | grep RES | grep -v total >> $OUT
KG_CONV_BS=1 KG_EXP_TAG=bsw timeout 300 python tools/exp_conv.py 2>&1 | grep RES | grep -v total >> $OUT
cat $OUT
for n in 64 192; do echo "== N=$n"; KG_LIB=build_ab/libkgan_bstiming.so KG_TIME_N=$n KG_TIME_CASES="D1 tail" KG_TIME_PLANS="bs" timeout 300 python tools/time_conv.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/bsw_stamps.log
