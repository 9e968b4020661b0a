#!/bin/bash
# round 6's final measurement visit (GPU box).  KG_COMMIT names the code state in the records.
#   tests + eager kernel stats + per-launch timeline + graph bench (tools/gpu_r06.sh), per-family kernel time json (bench.py's
#   work.family_rates), roofline PMC passes at 64 and 192 samples AND for the two C5a legs (kernel stats without --no-c5a),
#   the full default bench line, smoke(), the other BASELINE configurations, the fused-generator A/B
set -u
TAG=${1:-r06_final}
VARIANTS="staged:KG_GEN_FUSED=0 epilogue:KG_CONV_INKERNEL=0" bash tools/gpu_r06.sh $TAG
python tools/family_time.py gpurun_out/${TAG}_eager_kernel_stats.csv 6 ${KG_COMMIT:-unknown} > gpurun_out/${TAG}_eager_kernel_stats.json
mkdir -p profiles; cp gpurun_out/${TAG}_eager_kernel_stats.json profiles/${TAG}_eager_kernel_stats.json     # bench.py reads the newest r*_final_*
bash tools/roofline_pmc.sh 64 > gpurun_out/roofline_pmc.log 2>&1; tail -3 gpurun_out/roofline_pmc.log
bash tools/roofline_pmc.sh 192 > gpurun_out/roofline_pmc_bs192.log 2>&1; tail -3 gpurun_out/roofline_pmc_bs192.log
cp gpurun_out/roofline_bs64/roofline_pmc.json profiles/roofline_pmc.json; cp gpurun_out/roofline_bs192/roofline_pmc_bs192.json profiles/roofline_pmc_bs192.json
bash tools/roofline_c5a_pmc.sh > gpurun_out/roofline_c5a_pmc.log 2>&1; tail -3 gpurun_out/roofline_c5a_pmc.log
cp gpurun_out/roofline_c5a/roofline_c5a_pmc.json profiles/roofline_c5a_pmc.json
cp gpurun_out/roofline_c5a/stats/*kernel_stats.csv gpurun_out/${TAG}_roofline_c5a_kernel_stats.csv 2>/dev/null
( time python bench.py > gpurun_out/${TAG}_bench_default.json.log 2> gpurun_out/${TAG}_bench_default.err ) 2> gpurun_out/${TAG}_bench_default.time
tail -3 gpurun_out/${TAG}_bench_default.time
python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
: > gpurun_out/${TAG}_other_configs.log
for c in "stress 64" "ntu120 32" "h36m 64"; do set -- $c
  echo "== --config $1 --batch $2" >> gpurun_out/${TAG}_other_configs.log
  timeout 600 python bench.py --config $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-extras 2>&1 | tail -1 | cut -c1-700 >> gpurun_out/${TAG}_other_configs.log
done
cut -c1-260 gpurun_out/${TAG}_other_configs.log
