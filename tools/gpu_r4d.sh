#!/bin/bash
mkdir -p gpurun_out
for n in 64 192; do
KG_TIME_DUMP_CU=1 KG_LIB=build_ab/libkgan_p3timing.so KG_TIME_N=$n KG_TIME_CASES="D1 tail" KG_TIME_PLANS="2,1" timeout 300 python tools/time_conv.py 2>&1 | tail -16 | cut -c1-900
done
