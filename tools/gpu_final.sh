#!/bin/bash
# the round's final measurement visit (GPU box): tests + eager kernel stats + graph bench (tools/gpu_quick.sh), the
# full default bench line, smoke(), and the roofline PMC passes (bs=64 and the critic's bs=192 launch).
# KG_COMMIT names the code state in the records.   usage: gpu_final.sh <tag, e.g. r03_final>
set -u
TAG=${1:-r03_final}
bash tools/gpu_quick.sh $TAG
python tools/family_time.py gpurun_out/${TAG}_eager_kernel_stats.csv 6 ${KG_COMMIT:-unknown} > gpurun_out/${TAG}_eager_kernel_stats.json
mkdir -p profiles; cp gpurun_out/${TAG}_eager_kernel_stats.json profiles/${TAG}_eager_kernel_stats.json     # bench.py reads the newest r*_final_*
bash tools/roofline_pmc.sh 64 > gpurun_out/roofline_pmc.log 2>&1; tail -3 gpurun_out/roofline_pmc.log
bash tools/roofline_pmc.sh 192 > gpurun_out/roofline_pmc_bs192.log 2>&1; tail -3 gpurun_out/roofline_pmc_bs192.log
cp gpurun_out/roofline_bs64/roofline_pmc.json profiles/roofline_pmc.json; cp gpurun_out/roofline_bs192/roofline_pmc_bs192.json profiles/roofline_pmc_bs192.json
( time python bench.py > gpurun_out/${TAG}_bench_default.json.log 2> gpurun_out/${TAG}_bench_default.err ) 2> gpurun_out/${TAG}_bench_default.time
tail -3 gpurun_out/${TAG}_bench_default.time
python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
