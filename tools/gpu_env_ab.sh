#!/bin/bash
# whole-iteration A/B of one environment switch of the in-tree library, alternating runs:  AB="KG_CONV_PLAIN_EPI=0" bash tools/gpu_env_ab.sh
mkdir -p gpurun_out
OUT=gpurun_out/env_ab.log
: > $OUT
for i in 1 2 3; do
  for v in "$AB" "default"; do
    if [ "$v" = default ]; then e=""; else e="$v"; fi
    env $e timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-24s ms_per_step %.4f  samples/s %.0f' % ('$v', d['ms_per_step'], d['value']))" | tee -a $OUT
  done
done
