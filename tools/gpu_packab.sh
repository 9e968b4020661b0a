#!/bin/bash
# round 5: whole-iteration A/B of the cached packed tail weights (bench.py --no-pack-cache against the default), alternating runs
mkdir -p gpurun_out
OUT=gpurun_out/packab.log
: > $OUT
for i in 1 2 3; do
  for f in "--no-pack-cache" ""; do
    timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline $f 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-16s ms_per_step %.4f  samples/s %.0f' % ('$f' or 'pack-cache', d['ms_per_step'], d['value']))" | tee -a $OUT
  done
done
