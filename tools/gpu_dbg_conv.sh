#!/bin/bash
# kg_conv workgroup timeline + per-slice segments (instrumented build); KG_EXTRA_DEFS for experiment builds
set -u
mkdir -p gpurun_out
export KG_TIME_CASES="${KG_TIME_CASES:-D1 tail,D1 gcn}"
timeout 600 python tools/time_conv.py 2>&1 | grep -E "plan|setup|slice loop|epilogue|per slice|span|rror"
