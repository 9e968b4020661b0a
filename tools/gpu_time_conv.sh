#!/bin/bash
# GPU-box visit: workgroup timeline of kg_conv (instrumented build)
mkdir -p gpurun_out
timeout 600 python tools/time_conv.py > gpurun_out/time_conv.log 2>&1
tail -60 gpurun_out/time_conv.log
