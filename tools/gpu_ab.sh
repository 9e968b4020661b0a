#!/bin/bash
# GPU visit: A/B of an experiment build of kg_conv (built into /tmp on the box) against the in-tree library
set -u
mkdir -p gpurun_out
SRC=$(ls kinetic-gan_amd/csrc/*.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form ${KG_AB_FLAGS:-} -I include -I kinetic-gan_amd/csrc -o /tmp/libkgan_ab.so $SRC -ldl || exit 1
export KG_TUNE_QUICK=${KG_TUNE_QUICK:-64x128,32x128}
SCRIPT=${KG_AB_SCRIPT:-tools/tune_conv.py}
echo "== baseline" > gpurun_out/ab.log
timeout 600 python $SCRIPT >> gpurun_out/ab.log 2>&1
echo "== experiment ${KG_AB_FLAGS:-}" >> gpurun_out/ab.log
KG_LIB=/tmp/libkgan_ab.so timeout 600 python $SCRIPT >> gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
