#!/bin/bash
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_parity_gpu.py -q -m gpu --tb=short -p no:cacheprovider 2>&1 | tail -4
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>&1 | tail -1 | cut -c1-250
