#!/bin/bash
# experiment builds of the library with different minimum-waves launch bounds of kg_conv (KG_CONV_MINW32/64/128),
# whole-iteration time for each (the box's copy of the library is overwritten; nothing is committed)
set -u
mkdir -p gpurun_out
SRC=$(ls kinetic-gan_amd/csrc/*.hip)
for F in "" "-DKG_CONV_MINW64=4" "-DKG_CONV_MINW64=3" "-DKG_CONV_MINW32=6" "-DKG_CONV_MINW128=3"; do
  echo "== flags: '$F'"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form $F -I include -I kinetic-gan_amd/csrc -o kinetic-gan_amd/libkgan_hip.so $SRC -ldl 2>/dev/null || { echo build failed; continue; }
  PARTS=d_step,iteration timeout 300 python tools/time_parts.py 2>&1 | grep -E "d_step|iteration"
done
