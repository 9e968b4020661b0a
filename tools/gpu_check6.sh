#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 600 python tools/time_c5a.py > gpurun_out/time_c5a.log 2>&1; cat gpurun_out/time_c5a.log
bash tools/gpu_check2.sh
