#!/bin/bash
# round 5: SQ counters of the D1 tail at 192 samples, direct kernel vs a ring tile (RT)
export SHAPE=192,32,64,64,11,11,1 REPS=6
bash tools/pmc_kernel.sh tools/one_conv.py kg_conv_kernel base192
export KG_CONV_RING=1 KG_CONV_RING_TILE=${RT:-6}
bash tools/pmc_kernel.sh tools/one_conv.py kg_conv_ring ring192
cat gpurun_out/pmc_base192/summary.txt gpurun_out/pmc_ring192/summary.txt > gpurun_out/ring_pmc_summary.txt
