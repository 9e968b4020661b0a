#!/bin/bash
# round 5: tools/probe/presplit_probe.hip - the D1-tail contraction with operands that arrive pre-split (bf16 triples, k-interleaved)
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-result ${PROBE_FLAGS:-} -o /tmp/presplit_probe tools/probe/presplit_probe.hip > gpurun_out/presplit_build.log 2>&1 || { tail -5 gpurun_out/presplit_build.log; exit 1; }
/tmp/presplit_probe 2>&1 | tee gpurun_out/presplit_probe.log
