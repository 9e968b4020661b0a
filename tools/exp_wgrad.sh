#!/bin/bash
# GPU box: tools/exp_wgrad.py for the in-tree library and every build_ab variant in VARIANTS
set -u
echo "== base"; python tools/exp_wgrad.py 2>&1 | tail -8
for v in ${VARIANTS:-}; do
  echo "== $v"; KG_LIB=build_ab/libkgan_$v.so python tools/exp_wgrad.py 2>&1 | tail -8
done
