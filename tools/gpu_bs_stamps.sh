#!/bin/bash
# round 5: workgroup time stamps of the bf16-split tile kernel next to the direct kernel (tools/time_conv.py on a -DKG_CONV_TIMING build)
mkdir -p gpurun_out
for n in 64 192; do
echo "== N=$n"
KG_LIB=build_ab/libkgan_bstiming.so KG_TIME_N=$n KG_TIME_CASES="D1 tail" KG_TIME_PLANS="2,1;bs" timeout 300 python tools/time_conv.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/bs_stamps.log
