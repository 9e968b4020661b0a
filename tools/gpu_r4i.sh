#!/bin/bash
mkdir -p gpurun_out
for bc in 0 2048 4096 16384; do for bud in 2048 3072; do echo "== BIGCOLS=$bc BUDGET=$bud"; KG_WGRAD_BIGCOLS=$bc KG_WGRAD_BUDGET=$bud python tools/time_wgrad_many.py 2>&1 | grep -E "all 16|D4|D5"; done; done
