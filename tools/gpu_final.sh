#!/bin/bash
# the round's final measurement visit (GPU box): tests + eager kernel stats + graph bench (tools/gpu_quick.sh), the
# full default bench line, smoke(), and the roofline PMC passes.  KG_COMMIT names the code state in the records.
set -u
TAG=${1:-r02_final}
bash tools/gpu_quick.sh $TAG
python tools/family_time.py gpurun_out/${TAG}_eager_kernel_stats.csv 6 ${KG_COMMIT:-unknown} > gpurun_out/${TAG}_eager_kernel_stats.json
mkdir -p profiles; cp gpurun_out/${TAG}_eager_kernel_stats.json profiles/r02_final_eager_kernel_stats.json     # bench.py reads it
( time python bench.py > gpurun_out/${TAG}_bench_default.json.log 2> gpurun_out/${TAG}_bench_default.err ) 2> gpurun_out/${TAG}_bench_default.time
tail -3 gpurun_out/${TAG}_bench_default.time
python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
bash tools/roofline_pmc.sh > gpurun_out/roofline_pmc.log 2>&1; tail -3 gpurun_out/roofline_pmc.log
