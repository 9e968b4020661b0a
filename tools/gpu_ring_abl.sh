#!/bin/bash
# round 5: what bounds the ring form?  Ablation builds (wrong results on purpose; tools/build_variant.sh r_<X> "-DKG_RING_<X>")
# at a few shapes.   ABL_TILES="0 1" ABL_CASES="D1 tail,D3 gcn 128"
set -u
mkdir -p gpurun_out
OUT=gpurun_out/ring_abl.log
: > $OUT
export KG_EXP_N=${KG_EXP_N:-192}
export KG_EXP_CASES="${ABL_CASES:-D1 tail,D2 tail,D3 gcn 128}"
timeout 300 python tools/exp_conv.py >> $OUT 2>&1
for t in ${ABL_TILES:-0 1 2}; do
  KG_CONV_RING=1 KG_CONV_RING_TILE=$t KG_EXP_TAG=ring$t timeout 300 python tools/exp_conv.py >> $OUT 2>&1
  for v in ${ABL_VARIANTS:-NODMA NOBARRIER NOWAIT NOSTORE NOALL}; do
    KG_LIB=build_ab/libkgan_r_$v.so KG_CONV_RING=1 KG_CONV_RING_TILE=$t KG_EXP_TAG=ring$t-$v timeout 300 python tools/exp_conv.py >> $OUT 2>&1
  done
done
grep "^RES" $OUT | grep -v total | awk -F'|' '{printf "%-28s %-26s %s\n", $1, $2, $3}'
