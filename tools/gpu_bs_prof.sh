#!/bin/bash
# round 5: per-kernel durations of the bf16-split form on one exp_conv case (rocprofv3 --kernel-trace --stats)
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 64 192; do
KG_CONV_BS=1 KG_EXP_N=$n KG_EXP_CASES="${CASE:-D1 tail 64}" rocprofv3 --kernel-trace --stats -d /tmp/bsprof$n -o bs -- python3 $R/tools/exp_conv.py > /tmp/bsprof$n.log 2>&1
f=$(find /tmp/bsprof$n -name "*kernel_stats.csv" | head -1)
echo "== N=$n"; head -8 $f | cut -c1-220
done 2>&1 | tee $R/gpurun_out/bs_prof.log
