#!/bin/bash
# GPU visit: parity tests + A/B of the parameter-gradient sink (direct accumulation, side stream)
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/parity.log 2>&1
echo "parity rc=$?" >> gpurun_out/parity.log; tail -8 gpurun_out/parity.log
for mode in "sink+side" "sink" "autograd"; do
  case $mode in
    "sink+side") export KG_PARAM_SINK=1 KG_PARAM_SIDE_STREAM=1;;
    "sink")      export KG_PARAM_SINK=1 KG_PARAM_SIDE_STREAM=0;;
    "autograd")  export KG_PARAM_SINK=0 KG_PARAM_SIDE_STREAM=0;;
  esac
  timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/bench_$mode.log 2>&1
  echo "$mode graph: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_$mode.log)  $(tail -1 gpurun_out/bench_$mode.log | cut -c1-150 | grep -v metric)"
  timeout 400 python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-roofline > gpurun_out/bench_eager_$mode.log 2>&1
  echo "$mode eager: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_eager_$mode.log)"
done
