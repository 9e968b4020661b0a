#!/bin/bash
# round 5: the C5a aggregation launch (bench leg roofline_agg) on the kg_agg.hip of every round's end and of HEAD,
# each built as a mini library (kg_agg.hip + kg_misc.hip of that tree) with its own driver (tools/probe/agg_c5a_driver.cpp)
mkdir -p gpurun_out
for pass in 1 2; do
for t in ${TAGS:-r1 r2 r3 r4 head}; do
  timeout 120 ./build_ab/agg_c5a_$t build_ab/libagg_$t.so 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/agg_bisect.log
