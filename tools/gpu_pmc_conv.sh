#!/bin/bash
set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { cat gpurun_out/build.log; exit 1; }
SHAPE=192,128,256,32,5,5,2 bash tools/pmc_kernel.sh tools/one_conv.py kg_conv_kernel d3tail192
SHAPE=192,32,64,64,11,11,1 bash tools/pmc_kernel.sh tools/one_conv.py kg_conv_kernel d1tail192
cat gpurun_out/pmc_d3tail192/summary.txt gpurun_out/pmc_d1tail192/summary.txt > gpurun_out/pmc_conv_r02.txt
