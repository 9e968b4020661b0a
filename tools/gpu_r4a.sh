#!/bin/bash
# round-4 visit A: baseline bench line, kg_conv prologue experiments (A/B over the 13 D shapes), phase stamps
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/bench_base.log 2>&1; tail -1 gpurun_out/bench_base.log
VARIANTS="build_ab/libkgan_novmap.so build_ab/libkgan_vtab.so build_ab/libkgan_prio.so build_ab/libkgan_priovtab.so" bash tools/exp_conv.sh
for v in timing timingvtab; do
  for n in 64 192; do
    echo "== $v N=$n"
    KG_LIB=build_ab/libkgan_$v.so KG_TIME_N=$n KG_TIME_CASES="D1 tail" KG_TIME_PLANS="2,1" timeout 300 python tools/time_conv.py 2>&1 | tail -8
  done
done > gpurun_out/time_conv_r4a.log 2>&1
cat gpurun_out/time_conv_r4a.log
