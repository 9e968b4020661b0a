#!/bin/bash
set -u
mkdir -p gpurun_out
KG_LIB=build_ab/libkgan_p3.so timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv" 2>&1 | tail -3
VARIANTS="build_ab/libkgan_p3.so build_ab/libkgan_p3.so:KG_CONV_PERSIST_R=2 build_ab/libkgan_p3.so:KG_CONV_PERSIST_R=4 build_ab/libkgan_p3.so:KG_CONV_PERSIST_R=8 build_ab/libkgan_p3.so:KG_CONV_PERSIST_R=12" bash tools/exp_conv.sh
