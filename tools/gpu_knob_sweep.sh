#!/bin/bash
# end-of-round re-sweep of the existing tuning knobs against the whole iteration (one run each, a default run between every two)
mkdir -p gpurun_out
OUT=gpurun_out/knob_sweep.log
: > $OUT
run() { env $1 timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s ms_per_step %.4f' % ('${1:-default}', d['ms_per_step']))" | tee -a $OUT; }
run ""
for kv in KG_WGRAD_BUDGET=4096 KG_WGRAD_BUDGET=8192 KG_CONV_XCD_MIN=1000 KG_CONV_XCD_MIN=2500 KG_AGG_MFMA_GRID=768 KG_AGG_MFMA_GRID=384 KG_AGG_OUTER_BUDGET=768 KG_AGG_OUTER_BUDGET=384 KG_CONV_KW=0 KG_CONV_MANY=0 KG_CONV_INKERNEL_MAX=2 KG_CONV_INKERNEL_MAX=8 KG_GEN_FUSED_MINCOLS=32; do
  run $kv
  run ""
done
