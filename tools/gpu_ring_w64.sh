#!/bin/bash
mkdir -p gpurun_out
RT="10" bash tools/gpu_ringtest.sh
export KG_EXP_CASES="D1" KG_EXP_N=64,192
OUT=gpurun_out/ring_w64.log; : > $OUT
timeout 300 python tools/exp_conv.py >> $OUT 2>&1
for st in 0 16 32 48 64 96; do
  KG_CONV_RING=1 KG_CONV_RING_TILE=10 KG_CONV_RING_STAGGER=$st KG_EXP_TAG=w64-st$st timeout 300 python tools/exp_conv.py >> $OUT 2>&1
done
grep "^RES" $OUT | grep -v total | awk -F'|' '{printf "%-28s %-26s %s\n", $1, $2, $3}'
