#!/bin/bash
mkdir -p gpurun_out
OUT=gpurun_out/split_abl.log; : > $OUT
export KG_EXP_N=64,192 KG_EXP_CASES="D1 tail,D2 tail,D3 gcn 128,D3 tail"
KG_LIB=build_ab/libkgan_split_base.so KG_EXP_TAG=fp32 timeout 300 python tools/exp_conv.py >> $OUT 2>&1
for v in base halfx nocvt both; do
  KG_LIB=build_ab/libkgan_split_$v.so KG_CONV_SPLIT=1 KG_EXP_TAG=split-$v timeout 300 python tools/exp_conv.py >> $OUT 2>&1
done
grep "^RES" $OUT | grep -v total | awk -F'|' '{printf "%-24s %-24s %s\n", $1, $2, $3}'
