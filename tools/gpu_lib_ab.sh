#!/bin/bash
# whole-iteration A/B of two builds of the library (build_ab/libkgan_<tag>.so, tools/build_variant.sh), alternating runs:
#   LIBS="aggold aggnew" bash tools/gpu_lib_ab.sh
mkdir -p gpurun_out
OUT=gpurun_out/lib_ab.log
: > $OUT
for i in 1 2 3; do
  for t in $LIBS; do
    KG_LIB=build_ab/libkgan_$t.so timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-12s ms_per_step %.4f  samples/s %.0f' % ('$t', d['ms_per_step'], d['value']))" | tee -a $OUT
  done
done
