#!/bin/bash
set -u
mkdir -p gpurun_out
bash tools/gpu_check2.sh
