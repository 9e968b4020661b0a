#!/bin/bash
# build, then run the given command line: bash tools/gpu_run.sh python tools/xyz.py
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
"$@" 2>&1 | grep -v amdgpu.ids
