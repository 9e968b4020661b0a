#!/bin/bash
# GPU visit: kernel + parity tests (detail log), rocprofv3 kernel stats of the eager step, benches.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
rm -f gpurun_out/parity_detail.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log; tail -15 gpurun_out/kernels.log
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/parity.log
tail -15 gpurun_out/parity.log
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eager -o r01 -- python3 $R/bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/prof_eager.log 2>&1
echo "prof rc=$?"
cd $R
find gpurun_out/prof_eager -type f | head
find gpurun_out/prof_eager -type f ! -name "*stats*" -delete
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1; tail -2 gpurun_out/bench_graph.log
