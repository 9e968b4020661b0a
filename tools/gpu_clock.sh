#!/bin/bash
# effective shader clock under a saturated fp32 matrix pipe (tools/probe/clock_probe.hip)
set -u
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_probe tools/probe/clock_probe.hip 2>/dev/null || exit 1
/tmp/clock_probe
