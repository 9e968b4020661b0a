#!/bin/bash
# ring probe (tools/probe/ring_probe.hip) on the GPU box: RING_VARIANTS = list of flag sets separated by ';'
set -u
mkdir -p gpurun_out
IFS=';' read -ra VARS <<< "${RING_VARIANTS:- }"
for f in "${VARS[@]}"; do
  echo "== flags: $f"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form $f -o /tmp/ring_probe tools/probe/ring_probe.hip 2>/dev/null || { echo build failed; continue; }
  for n in ${RING_COLS:-131072 32768 1048576}; do timeout 60 /tmp/ring_probe $n | grep "grid 256"; done
done
