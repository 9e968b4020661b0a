#!/bin/bash
# hipGraph-replayed time of the iteration's parts (tools/time_parts.py), generator trunk on / off
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
echo "== generator trunk"; KG_GEN_TRUNK=1 timeout 300 python tools/time_parts.py 2>&1 | grep -v amdgpu.ids
echo "== generator block-wise"; KG_GEN_TRUNK=0 timeout 300 python tools/time_parts.py 2>&1 | grep -v amdgpu.ids
