#!/bin/bash
# One GPU-box visit: kernel tests, parity tests, smoke, bench.  Logs go to gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/gpu.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > gpurun_out/cpu.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x --tb=short -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log
tail -25 gpurun_out/kernels.log
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/parity.log
tail -40 gpurun_out/parity.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-graph > gpurun_out/bench_eager.log 2>&1; echo "rc=$?" >> gpurun_out/bench_eager.log; tail -5 gpurun_out/bench_eager.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1; echo "rc=$?" >> gpurun_out/bench_graph.log; tail -5 gpurun_out/bench_graph.log
