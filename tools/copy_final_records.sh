#!/bin/bash
# run HERE after a tools/gpu_r06_final.sh visit: copies the visit's records from gpurun_out/ into profiles/ (copies made on the
# GPU box do not come back: only gpurun_out/ is merged)
set -eu
TAG=${1:-r06_final}
cd "$(dirname "$0")/.."
G=gpurun_out
cp $G/${TAG}_bench_default.json.log $G/${TAG}_eager_kernel_stats.csv $G/${TAG}_eager_kernel_stats.json $G/${TAG}_last_step.txt \
   $G/${TAG}_other_configs.log $G/${TAG}_smoke.log $G/${TAG}_roofline_c5a_kernel_stats.csv profiles/
cp $G/roofline_bs64/roofline_pmc.json profiles/roofline_pmc.json
cp $G/roofline_bs192/roofline_pmc_bs192.json profiles/roofline_pmc_bs192.json
cp $G/roofline_c5a/roofline_c5a_pmc.json profiles/roofline_c5a_pmc.json
cp $G/roofline_bs64/stats/rf_kernel_stats.csv profiles/${TAG}_roofline_bs64_kernel_stats.csv
cp $G/roofline_bs192/stats/rf_kernel_stats.csv profiles/${TAG}_roofline_bs192_kernel_stats.csv
for v in default staged epilogue; do for r in 1 2; do
  echo "$v $r: $(tail -1 $G/bench_${TAG}_${v}_$r.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done; done > profiles/${TAG}_genblock_ab.log
cat profiles/${TAG}_genblock_ab.log
