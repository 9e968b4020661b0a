#!/bin/bash
# round 5: the transposed-accumulator epilogue (16-byte stores) of the bf16-split tile kernel: tests, then timing with KG_CONV_STORE_T=0 / default
# (the transposed epilogue and -DKG_BS_NOPACK are NOT in the tree: tools/probe/conv_bs_ablation_switches.patch)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv and (bs0 or bs1 or bs2)" 2>&1 | tail -5 | tee gpurun_out/bs_tests.log
OUT=gpurun_out/bs_t.log
: > $OUT
for st in 0 1; do
KG_CONV_STORE_T=$st KG_LIB=build_ab/libkgan_bsnopack.so KG_CONV_BS=1 KG_EXP_TAG=bs-nopack-T$st timeout 300 python tools/exp_conv.py 2>&1 | grep RES | grep -v total >> $OUT
done
cat $OUT
for st in 0 1; do echo "== KG_CONV_STORE_T=$st"; KG_CONV_STORE_T=$st KG_LIB=build_ab/libkgan_bstiming.so KG_TIME_N=64 KG_TIME_CASES="D1 tail" KG_TIME_PLANS="bs" timeout 300 python tools/time_conv.py 2>&1 | grep -v amdgpu.ids; done | tee -a $OUT
