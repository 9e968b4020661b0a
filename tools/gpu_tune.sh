#!/bin/bash
# kg_conv plan tuning at 64 and 192 samples (tools/tune_conv.py)
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -5 gpurun_out/build.log; exit 1; }
for n in 64 192; do
  N=$n timeout 900 python tools/tune_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${1:-r03}_tune_conv_n$n.log
done
