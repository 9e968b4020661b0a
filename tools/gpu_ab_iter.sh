#!/bin/bash
# GPU box: whole-iteration ms (hipGraph) for build_ab variants, interleaved REPS times.  VARIANTS="a b ..." (base = in-tree)
set -u
for r in $(seq ${REPS:-2}); do
for v in $VARIANTS; do
  if [ $v = base ]; then unset KG_LIB; else export KG_LIB=build_ab/libkgan_$v.so; fi
  echo -n "$v: "
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])[\"ms_per_step\"])"
done; done
