#!/bin/bash
# GPU box: kg_wgrad variants (build_ab/libkgan_<v>.so) - in-step launch times, whole-iteration bench, wgrad tests
set -u
VARIANTS="$VARIANTS" bash tools/gpu_wgprof.sh 2>&1 | grep -v "reduce\|^base" 
for v in $VARIANTS; do
  echo "== $v"
  KG_LIB=build_ab/libkgan_$v.so python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])[\"ms_per_step\"])"
  KG_LIB=build_ab/libkgan_$v.so timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k wgrad -p no:cacheprovider 2>&1 | tail -1
done
