#!/bin/bash
# round-4 visit B: new kg_conv prologue vs the round-3 one: A/B over the 13 D shapes, phase stamps, kernel tests
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
VARIANTS="build_ab/libkgan_base0.so build_ab/libkgan_p1.so build_ab/libkgan_p1prio.so" bash tools/exp_conv.sh
for v in base0timing p1timing; do
  for n in 64 192; do
    echo "== $v N=$n"
    KG_LIB=build_ab/libkgan_$v.so KG_TIME_N=$n KG_TIME_CASES="D1 tail" KG_TIME_PLANS="2,1" timeout 300 python tools/time_conv.py 2>&1 | tail -8
  done
done > gpurun_out/time_conv_r4b.log 2>&1
cat gpurun_out/time_conv_r4b.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv" 2>&1 | tail -5
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>&1 | tail -1 | cut -c1-400
