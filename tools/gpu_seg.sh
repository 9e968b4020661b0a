#!/bin/bash
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
for F in "" "--segmented --no-overlap" "--segmented"; do
  echo "== bench $F"
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras $F 2>&1 | grep -v "amdgpu.ids\|Warn\|run_backward" | tail -1 | cut -c1-330
done
