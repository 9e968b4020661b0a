#!/bin/bash
# experiment builds of the library with different minimum-waves launch bounds of kg_conv (KG_CONV_MINW32/64/128),
# whole-iteration time for each (built into /tmp and selected with KG_LIB: the in-tree library is never touched)
set -u
mkdir -p gpurun_out
SRC=$(ls kinetic-gan_amd/csrc/*.hip)
for F in "" "-DKG_CONV_MINW64=4" "-DKG_CONV_MINW64=3" "-DKG_CONV_MINW32=6" "-DKG_CONV_MINW128=3"; do
  echo "== flags: '$F'"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form $F -I include -I kinetic-gan_amd/csrc -o /tmp/libkgan_minw.so $SRC -ldl 2>/dev/null || { echo build failed; continue; }
  KG_LIB=/tmp/libkgan_minw.so PARTS=d_step,iteration timeout 300 python tools/time_parts.py 2>&1 | grep -E "d_step|iteration"
done
