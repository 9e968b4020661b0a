#!/bin/bash
# quick visit: build, parity tests, eager kernel stats + last-step timeline, hipGraph bench line
set -u
TAG=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
rm -f gpurun_out/parity_detail.log
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log; tail -5 gpurun_out/kernels.log
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/parity.log; tail -15 gpurun_out/parity.log
fi
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eager -o $TAG -- python3 $R/bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/prof_eager.log 2>&1
echo "prof rc=$?"
cd $R
python tools/last_step.py gpurun_out/prof_eager/${TAG}_kernel_trace.csv > gpurun_out/${TAG}_last_step.txt 2> gpurun_out/last_step.err; tail -3 gpurun_out/${TAG}_last_step.txt
cp gpurun_out/prof_eager/${TAG}_kernel_stats.csv gpurun_out/${TAG}_eager_kernel_stats.csv
find gpurun_out/prof_eager -type f ! -name "*stats*" -delete
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/bench_graph.log 2>&1; tail -1 gpurun_out/bench_graph.log
