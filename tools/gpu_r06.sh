#!/bin/bash
# round 6 visit: (optional) tests, eager kernel stats + last-step timeline, hipGraph bench line, for the default
# configuration and for every "NAME:ENV=VAL[,ENV=VAL]" variant in VARIANTS (bench line only); KTESTS="-k word" or
# KSEL="expr with spaces" select kernel tests
set -u
TAG=${1:-r06}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x ${KTESTS:-} ${KSEL:+-k "$KSEL"} > gpurun_out/kernels.log 2>&1
echo "kernels rc=$?" >> gpurun_out/kernels.log; tail -5 gpurun_out/kernels.log
rm -f gpurun_out/parity_detail.log
[ "${SKIP_PARITY:-0}" = "1" ] || timeout 1200 python -m pytest tests/test_parity_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/parity.log; tail -15 gpurun_out/parity.log
fi
R=$GRAFT_REPO_ROOT
if [ "${SKIP_PROF:-0}" != "1" ]; then
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eager -o $TAG -- python3 $R/bench.py --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-roofline --no-extras > $R/gpurun_out/prof_eager.log 2>&1
echo "prof rc=$?"
cd $R
python tools/last_step.py gpurun_out/prof_eager/${TAG}_kernel_trace.csv > gpurun_out/${TAG}_last_step.txt 2> gpurun_out/last_step.err; tail -3 gpurun_out/${TAG}_last_step.txt
cp gpurun_out/prof_eager/${TAG}_kernel_stats.csv gpurun_out/${TAG}_eager_kernel_stats.csv
find gpurun_out/prof_eager -type f ! -name "*stats*" -delete
fi
B="--steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras"
for rep in 1 2; do
timeout 600 python bench.py $B > gpurun_out/bench_${TAG}_default_$rep.log 2>&1; echo "default: $(tail -1 gpurun_out/bench_${TAG}_default_$rep.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
for v in ${VARIANTS:-}; do
  name=${v%%:*}; envs=${v#*:}
  env $(echo $envs | tr ',' ' ') timeout 600 python bench.py $B > gpurun_out/bench_${TAG}_${name}_$rep.log 2>&1
  echo "$name ($envs): $(tail -1 gpurun_out/bench_${TAG}_${name}_$rep.log | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
done
