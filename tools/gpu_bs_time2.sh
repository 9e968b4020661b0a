#!/bin/bash
# round 5: the bf16-split form's tile kernel alone (variant built with -DKG_BS_NOPACK: no weight-pack launch) next to the full call
# (-DKG_BS_NOPACK is NOT in the tree: apply tools/probe/conv_bs_ablation_switches.patch, tools/build_variant.sh bsnopack "-DKG_BS_NOPACK" kg_conv.hip, revert)
mkdir -p gpurun_out
OUT=gpurun_out/bs_time2.log
: > $OUT
KG_EXP_TAG=direct timeout 300 python tools/exp_conv.py 2>&1 | grep RES >> $OUT
KG_CONV_BS=1 KG_EXP_TAG=bs timeout 300 python tools/exp_conv.py 2>&1 | grep RES >> $OUT
KG_LIB=build_ab/libkgan_bsnopack.so KG_CONV_BS=1 KG_EXP_TAG=bs-nopack timeout 300 python tools/exp_conv.py 2>&1 | grep RES >> $OUT
cat $OUT
