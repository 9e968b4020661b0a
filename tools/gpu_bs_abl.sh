#!/bin/bash
# round 5: ablation builds of the bf16-split tile kernel (timing only: -DKG_BS_ABL bits 1 no MFMA, 2 no split arithmetic, 4 no stores, 8 no feature loads,
# (the -DKG_BS_ABL / -DKG_BS_NOPACK switches are NOT in the tree: apply tools/probe/conv_bs_ablation_switches.patch, build the variants
#  with tools/build_variant.sh bsabl<bits> "-DKG_BS_NOPACK -DKG_BS_ABL=<bits>" kg_conv.hip, revert)
# 16 no fragment reads, 32 no weight path, 64 no feature stash, 128 return after the prologue, 512 return at entry)
mkdir -p gpurun_out
OUT=gpurun_out/bs_abl.log
: > $OUT
for v in ${ABLS:-nopack abl7 abl519 abl135 abl23 abl39 abl71 abl15 abl127}; do
KG_LIB=build_ab/libkgan_bs$v.so KG_CONV_BS=1 KG_EXP_TAG=bs-$v timeout 300 python tools/exp_conv.py 2>&1 | grep RES | grep -v total >> $OUT
done
cat $OUT
