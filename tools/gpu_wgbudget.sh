#!/bin/bash
# GPU box: KG_WGRAD_BUDGET sweep for one variant library (whole-iteration ms, hipGraph)
set -u
V=${1:-wg_single}
for b in ${BUDGETS:-2048 3072 4096 5120 6144 8192}; do
  echo -n "$V budget $b: "
  KG_WGRAD_BUDGET=$b KG_LIB=build_ab/libkgan_$V.so python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])[\"ms_per_step\"])"
done
