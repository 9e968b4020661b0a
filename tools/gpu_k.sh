#!/bin/bash
# run selected GPU kernel tests: bash tools/gpu_k.sh "<pytest -k expression>"
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "$1" 2>&1 | tail -15
